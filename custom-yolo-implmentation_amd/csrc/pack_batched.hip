// All conv weight matrices of a model packed by ONE launch per step (forward and data-gradient forms):
// the per-layer k_pack_weights launches were 176 of the ~1500 kernels of a training step, ~5 us each.
// A job table (device memory, built once per model) lists source parameter, destination, shape and taps;
// each thread finds its job by binary search on the element prefix.
#include "common.h"
#include "conv_geom.h"

struct PackJob {
    const void* w;     // OIHW parameter
    void* out;         // packed destination (rows x Kpad)
    long start;        // first global element index of this job
    long cstart;       // first 4096-element chunk (= workgroup) of this job
    int O, I, k, mode, ntaps, Kpad, rows, w_dtype;
    int kh[9], kw[9];
};

namespace {

template <typename P> __device__ __forceinline__ float ldw(const void* p, long i) { return to_f<P>(((const P*)p)[i]); }

constexpr int CHUNK = 4096;

// one workgroup = one 4096-element chunk of one job (job found once per workgroup, not once per element)
template <typename T>
__global__ void k_pack_batched(const PackJob* __restrict__ jobs, int njobs) {
    __shared__ int sj;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {                       // last job with cstart <= blockIdx.x
            int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].cstart <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sj = lo;
    }
    __syncthreads();
    const PackJob j = jobs[sj];
    const long nelem = (long)j.rows * j.Kpad;
    const long base = ((long)blockIdx.x - j.cstart) * CHUNK;
    const int inner = j.mode == 0 ? j.I : j.O;
    for (int t0 = threadIdx.x; t0 < CHUNK; t0 += blockDim.x) {
        const long le = base + t0;
        if (le >= nelem) break;
        const int r = (int)(le / j.Kpad), kk = (int)(le - (long)r * j.Kpad);
        float v = 0.f;
        if (kk < j.ntaps * inner) {
            const int t = kk / inner, c = kk - t * inner;
            const int o = j.mode == 0 ? r : c, i = j.mode == 0 ? c : r;
            const long src = (((long)o * j.I + i) * j.k + j.kh[t]) * j.k + j.kw[t];
            v = j.w_dtype == YOLO_F32 ? ldw<float>(j.w, src) : j.w_dtype == YOLO_BF16 ? ldw<bf16_t>(j.w, src) : ldw<f16_t>(j.w, src);
        }
        ((T*)j.out)[le] = from_f<T>(v);
    }
}

}  // namespace

extern "C" {

int yolo_pack_job_bytes(void) { return (int)sizeof(PackJob); }

// number of jobs one conv contributes for `mode` (dgrad of a stride-2 conv = 4 parity classes)
int yolo_pack_job_count(int stride, int mode) { return (mode == 1 && stride == 2) ? 4 : 1; }

// Fill the host-side job record(s) of one conv at `jobs_host` (yolo_pack_job_count records).  `out` is the base of this
// conv's packed buffer (same layout yolo_conv_pack_weights produces).  Returns the element count appended.
long yolo_pack_job_fill(void* jobs_host, const void* w, int w_dtype, void* out, int out_elem_bytes, int O, int I, int k,
                        int stride, int mode, long start) {
    PackJob* jobs = (PackJob*)jobs_host;
    int ncls = yolo_pack_job_count(stride, mode);
    long off = 0;
    for (int c = 0; c < ncls; ++c) {
        PackJob& j = jobs[c];
        int dh[9], dw[9];
        j.ntaps = conv_taps(mode, k, stride, c, dh, dw, j.kh, j.kw);
        j.w = w; j.w_dtype = w_dtype; j.O = O; j.I = I; j.k = k; j.mode = mode;
        j.rows = mode == 0 ? O : I;
        j.Kpad = round_up32(j.ntaps * (mode == 0 ? I : O));
        j.out = (char*)out + off * out_elem_bytes;
        j.start = start + off;
        off += (long)j.rows * j.Kpad;
    }
    return off;
}

// second pass over the host job table: assigns each job its first chunk; returns the total chunk (= workgroup) count
long yolo_pack_jobs_finalize(void* jobs_host, int njobs) {
    PackJob* jobs = (PackJob*)jobs_host;
    long c = 0;
    for (int i = 0; i < njobs; ++i) {
        jobs[i].cstart = c;
        c += ((long)jobs[i].rows * jobs[i].Kpad + CHUNK - 1) / CHUNK;
    }
    return c;
}

int yolo_pack_batched(const void* jobs_dev, int njobs, long nchunks, int out_dtype, hipStream_t st) {
    if (njobs <= 0 || nchunks <= 0) return YOLO_OK;
    YOLO_DISPATCH_T(out_dtype, hipLaunchKernelGGL((k_pack_batched<T>), dim3((unsigned)nchunks), dim3(256), 0, st, (const PackJob*)jobs_dev, njobs));
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
