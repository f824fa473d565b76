// AdamW step of ALL parameters in one launch.
// torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps added to sqrt(v)/sqrt(bc2)):
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// A device job table lists (param, grad, exp_avg, exp_avg_sq, numel) per tensor; one workgroup owns one 4096-element
// chunk of one tensor (job found once per workgroup).  Hyper-parameters and the step counter live in device memory so
// a captured step follows a learning-rate schedule without recapture; GradScaler's scale / found_inf pair is
// honoured like torch's fused optimizers do (skip the whole step on overflow, unscale the gradient in flight).
#include "common.h"

struct AdamJob {
    void* p;            // parameter, p_dtype
    const void* g;      // gradient, g_dtype
    float* m;           // exp_avg (fp32)
    float* v;           // exp_avg_sq (fp32)
    long n;             // elements
    long cstart;        // first 4096-element chunk (= workgroup) of this job
    int p_dtype, g_dtype;
};

namespace {

constexpr int CHUNK = 4096;

template <typename P> __device__ __forceinline__ float ldf(const void* p, long i) { return to_f<P>(((const P*)p)[i]); }
__device__ __forceinline__ float ld_any(const void* p, int dt, long i) {
    return dt == YOLO_F32 ? ldf<float>(p, i) : dt == YOLO_BF16 ? ldf<bf16_t>(p, i) : ldf<f16_t>(p, i);
}
__device__ __forceinline__ void st_any(void* p, int dt, long i, float x) {
    if (dt == YOLO_F32) ((float*)p)[i] = x;
    else if (dt == YOLO_BF16) ((bf16_t*)p)[i] = from_f<bf16_t>(x);
    else ((f16_t*)p)[i] = from_f<f16_t>(x);
}

// The update of ONE element, shared by the packet path and the any-dtype path: explicit fmaf / mul sequence, so that the
// same (p, g, m, v) give the same bits whichever path a tensor takes (a low-precision gradient of an fp32 master goes
// through the generic one; left to the compiler the two loops were contracted differently: 1-ulp differences between
// the sharded runner and torch FSDP2 on the same gradients).
__device__ __forceinline__ void adam_elem(float& p, float gg, float& m, float& v, float b1, float omb1, float b2, float omb2,
                                          float step_size, float bc2s, float decay, float eps) {
    m = __fmaf_rn(b1, m, __fmul_rn(omb1, gg));
    v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(omb2, gg), gg));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2s), eps);
    p = __fsub_rn(__fmul_rn(p, decay), __fdiv_rn(__fmul_rn(step_size, m), denom));
}

// step += 1 unless the scaler found an overflow (one thread; the main kernel reads the updated value)
__global__ void k_adamw_tick(float* __restrict__ step, const float* __restrict__ found_inf) {
    if (found_inf == nullptr || *found_inf == 0.f) *step += 1.f;
}

// hyper = [lr, beta1, beta2, eps, weight_decay] as DOUBLES: 1 - beta and the bias corrections are formed in double
// like torch does on the host (1 - 0.999 in fp32 is off by 5e-5 relative, which shows in exp_avg_sq)
__global__ __launch_bounds__(256) void k_adamw(const AdamJob* __restrict__ jobs, int njobs, const double* __restrict__ hyper,
                                               const float* __restrict__ step, const float* __restrict__ grad_scale,
                                               const float* __restrict__ found_inf) {
    if (found_inf != nullptr && *found_inf != 0.f) return;
    __shared__ int sj;
    __shared__ float sc[7];                     // b1, 1-b1, b2, 1-b2, step_size, 1/sqrt(bc2), decay
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {                       // last job with cstart <= blockIdx.x
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].cstart <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sj = lo;
        const double lr = hyper[0], b1d = hyper[1], b2d = hyper[2], wd = hyper[4], t = (double)*step;
        const double bc1 = 1.0 - pow(b1d, t), bc2 = 1.0 - pow(b2d, t);
        sc[0] = (float)b1d; sc[1] = (float)(1.0 - b1d); sc[2] = (float)b2d; sc[3] = (float)(1.0 - b2d);
        sc[4] = (float)(lr / bc1); sc[5] = (float)sqrt(bc2); sc[6] = (float)(1.0 - lr * wd);
    }
    __syncthreads();
    const AdamJob j = jobs[sj];
    const float b1 = sc[0], omb1 = sc[1], b2 = sc[2], omb2 = sc[3], step_size = sc[4], bc2s = sc[5], decay = sc[6];
    const float eps = (float)hyper[3];
    const float gs = grad_scale ? 1.f / *grad_scale : 1.f;      // GradScaler: gradients arrive multiplied by the scale
    const long base = ((long)blockIdx.x - j.cstart) * CHUNK;
    if (j.p_dtype == YOLO_F32 && j.g_dtype == YOLO_F32 && (j.n & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(j.p) | reinterpret_cast<uintptr_t>(j.g) | reinterpret_cast<uintptr_t>(j.m) |
          reinterpret_cast<uintptr_t>(j.v)) & 15) == 0) {
        // the common case: fp32 master weights and gradients, 16-byte packets
        for (long i = base + threadIdx.x * 4L; i < base + CHUNK && i < j.n; i += 256 * 4) {
            float4 p = *reinterpret_cast<float4*>((float*)j.p + i);
            const float4 g = *reinterpret_cast<const float4*>((const float*)j.g + i);
            float4 m = *reinterpret_cast<float4*>(j.m + i), v = *reinterpret_cast<float4*>(j.v + i);
            float* pp = &p.x; const float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) adam_elem(pp[k], gp[k] * gs, mp[k], vp[k], b1, omb1, b2, omb2, step_size, bc2s, decay, eps);
            *reinterpret_cast<float4*>((float*)j.p + i) = p;
            *reinterpret_cast<float4*>(j.m + i) = m;
            *reinterpret_cast<float4*>(j.v + i) = v;
        }
        return;
    }
    for (long i = base + threadIdx.x; i < base + CHUNK && i < j.n; i += 256) {
        float pv = ld_any(j.p, j.p_dtype, i), m = j.m[i], v = j.v[i];
        adam_elem(pv, ld_any(j.g, j.g_dtype, i) * gs, m, v, b1, omb1, b2, omb2, step_size, bc2s, decay, eps);
        j.m[i] = m;
        j.v[i] = v;
        st_any(j.p, j.p_dtype, i, pv);
    }
}

// ---- fp16 loss scaling on the device (torch.amp.GradScaler's protocol, reference src/training/train_model.py:195-208,
// 247-253, without its host round trips): found_inf over every gradient of the job table, then the optimizer kernel above
// (which skips on found_inf and unscales in flight), then the scale update -- all launches of a captured step.
__global__ __launch_bounds__(256) void k_found_inf(const AdamJob* __restrict__ jobs, int njobs, float* __restrict__ found_inf) {
    __shared__ int sj;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].cstart <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sj = lo;
    }
    __syncthreads();
    const AdamJob j = jobs[sj];
    const long base = ((long)blockIdx.x - j.cstart) * CHUNK;
    bool bad = false;
    for (long i = base + threadIdx.x; i < base + CHUNK && i < j.n; i += 256) {
        const float g = ld_any(j.g, j.g_dtype, i);
        bad |= !(fabsf(g) <= 3.4028234664e38f);             // inf or nan
    }
    if (bad) *found_inf = 1.f;                                // every writer stores the same value
}

// state = [scale, found_inf, last_found_inf] (fp32), tracker = successful steps since the last change.
// torch's _amp_update_scale_: overflow -> scale *= backoff, tracker = 0; else tracker + 1 == interval -> scale *= growth
// (if still finite), tracker = 0; else tracker += 1.  found_inf is handed on as last_found_inf and cleared for the next step.
__global__ void k_amp_update(float* __restrict__ state, int* __restrict__ tracker, float growth, float backoff, int interval) {
    const float found = state[1];
    if (found != 0.f) {
        state[0] *= backoff;
        *tracker = 0;
    } else {
        const int ok = *tracker + 1;
        if (ok == interval) {
            const float ns = state[0] * growth;
            if (fabsf(ns) <= 3.4028234664e38f) state[0] = ns;
            *tracker = 0;
        } else {
            *tracker = ok;
        }
    }
    state[2] = found;
    state[1] = 0.f;
}

}  // namespace

extern "C" {

int yolo_adamw_job_bytes(void) { return (int)sizeof(AdamJob); }

// write record `index` of the host-side table
int yolo_adamw_job_fill(void* jobs_host, int index, void* p, int p_dtype, const void* g, int g_dtype, float* m, float* v,
                        long n) {
    if (n < 0) return YOLO_ERR_ARG;
    AdamJob& j = ((AdamJob*)jobs_host)[index];
    j.p = p; j.g = g; j.m = m; j.v = v; j.n = n; j.cstart = 0; j.p_dtype = p_dtype; j.g_dtype = g_dtype;
    return YOLO_OK;
}

// only the gradient pointers of the n records changed (an eager loop's zero_grad(set_to_none=True) hands autograd fresh
// gradient tensors every step): one call instead of n yolo_adamw_job_fill calls
int yolo_adamw_jobs_set_grads(void* jobs_host, int njobs, const void* const* grads) {
    AdamJob* jobs = (AdamJob*)jobs_host;
    for (int i = 0; i < njobs; ++i) jobs[i].g = grads[i];
    return YOLO_OK;
}

// assign the chunk ranges; returns the number of workgroups
long yolo_adamw_jobs_finalize(void* jobs_host, int njobs) {
    AdamJob* jobs = (AdamJob*)jobs_host;
    long c = 0;
    for (int i = 0; i < njobs; ++i) {
        jobs[i].cstart = c;
        long nch = (jobs[i].n + CHUNK - 1) / CHUNK;
        c += nch > 0 ? nch : 1;
    }
    return c;
}

int yolo_adamw_step(const void* jobs_dev, int njobs, long nchunks, const double* hyper, float* step, const float* grad_scale,
                    const float* found_inf, hipStream_t st) {
    if (njobs <= 0 || nchunks <= 0) return YOLO_OK;
    hipLaunchKernelGGL(k_adamw_tick, dim3(1), dim3(1), 0, st, step, found_inf);
    hipLaunchKernelGGL(k_adamw, dim3((unsigned)nchunks), dim3(256), 0, st, (const AdamJob*)jobs_dev, njobs, hyper, step,
                       grad_scale, found_inf);
    return YOLO_LAUNCH_CHECK();
}

// One optimizer step under dynamic loss scaling, all on the device: amp_state = [scale, found_inf (0 on entry), last_found_inf].
// The gradients of the job table carry the factor `scale` (the loss kernel multiplied its gradient by it in fp32).
int yolo_adamw_amp_step(const void* jobs_dev, int njobs, long nchunks, const double* hyper, float* step, float* amp_state,
                        int* growth_tracker, float growth_factor, float backoff_factor, int growth_interval, hipStream_t st) {
    if (njobs <= 0 || nchunks <= 0) return YOLO_OK;
    if (!(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f) || growth_interval < 1) return YOLO_ERR_ARG;
    hipLaunchKernelGGL(k_found_inf, dim3((unsigned)nchunks), dim3(256), 0, st, (const AdamJob*)jobs_dev, njobs, amp_state + 1);
    hipLaunchKernelGGL(k_adamw_tick, dim3(1), dim3(1), 0, st, step, amp_state + 1);
    hipLaunchKernelGGL(k_adamw, dim3((unsigned)nchunks), dim3(256), 0, st, (const AdamJob*)jobs_dev, njobs, hyper, step,
                       amp_state, amp_state + 1);
    hipLaunchKernelGGL(k_amp_update, dim3(1), dim3(1), 0, st, amp_state, growth_tracker, growth_factor, backoff_factor, growth_interval);
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
