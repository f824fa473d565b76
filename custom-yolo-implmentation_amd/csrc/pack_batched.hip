// All conv weight matrices of a model packed by ONE launch per step (forward and data-gradient forms):
// the per-layer k_pack_weights launches were 176 of the ~1500 kernels of a training step, ~5 us each.
// A job table (device memory, built once per model) lists source parameter, destination, shape and taps.
// A workgroup owns a 32 x 32 block of (o, i) of ONE parameter for all k*k taps: it reads the OIHW block once
// (32 contiguous runs of 32*k*k values), keeps it in LDS and emits every packed form of that conv -- the forward
// matrix [o][tap][i] and the data-gradient matrices [i][tap][o] (four parity classes for stride 2) -- as 16-byte
// stores of 8 consecutive elements.  (The first form walked the packed elements one by one: a 4-byte gather with a
// stride of I*k*k floats per lane for the data-gradient forms and 2-byte stores, 141 us per step for 19 MB.)
#include "common.h"
#include "conv_geom.h"

struct PackJob {
    const void* w;     // OIHW parameter
    void* out;         // packed destination (rows x Kpad)
    long start;        // first global element index of this job
    long cstart;       // first tile (= workgroup) of this job's group
    int O, I, k, mode, ntaps, Kpad, rows, w_dtype;
    int kh[9], kw[9];
    int lead, ngrp;    // group = consecutive jobs of one parameter (forward form first): index of its first job, job count
};

namespace {

template <typename P> __device__ __forceinline__ float ldw(const void* p, long i) { return to_f<P>(((const P*)p)[i]); }

constexpr int TB = 32;            // tile edge in o and in i
constexpr int MAXGRP = 5;         // forward + up to four data-gradient classes

template <typename T>
__global__ __launch_bounds__(256) void k_pack_tiles(const PackJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[TB][TB * 9 + 1];
    __shared__ int sj;
    __shared__ int stap[MAXGRP][9];
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {                       // last job with cstart <= blockIdx.x
            int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].cstart <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sj = jobs[lo].lead;
    }
    __syncthreads();
    const int lead = sj;
    const PackJob& L = jobs[lead];
    const int O = L.O, I = L.I, k = L.k, kk = k * k, ngrp = L.ngrp, wdt = L.w_dtype;
    const void* w = L.w;
    for (int t = threadIdx.x; t < ngrp * 9; t += 256) {
        const PackJob& j = jobs[lead + t / 9];
        const int tt = t % 9;
        stap[t / 9][tt] = tt < j.ntaps ? j.kh[tt] * k + j.kw[tt] : 0;
    }
    const int nti = (I + TB - 1) / TB;
    const int tl = (int)((long)blockIdx.x - L.cstart);
    const int o0 = (tl / nti) * TB, i0 = (tl % nti) * TB;
    const int on = O - o0 < TB ? O - o0 : TB, in = I - i0 < TB ? I - i0 : TB;
    const int run = in * kk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wdt == YOLO_F32 && ((I * kk) & 3) == 0 && (run & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        // 16-byte loads, all of a thread's (up to 9) in flight before the first LDS store
        const int r4 = run >> 2, n4 = on * r4;
        float4 v[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int e = threadIdx.x + q * 256;
            if (e < n4) {
                const int r = e / r4, c = e - r * r4;
                v[q] = *reinterpret_cast<const float4*>((const float*)w + ((long)(o0 + r) * I + i0) * kk + c * 4);
            }
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int e = threadIdx.x + q * 256;
            if (e < n4) {
                const int r = e / r4, c = (e - r * r4) * 4;
                tile[r][c] = v[q].x; tile[r][c + 1] = v[q].y; tile[r][c + 2] = v[q].z; tile[r][c + 3] = v[q].w;
            }
        }
    } else
    for (int r = wave; r < on; r += 4) {
        const long src = ((long)(o0 + r) * I + i0) * kk;
        for (int c = lane; c < run; c += 64)
            tile[r][c] = wdt == YOLO_F32 ? ldw<float>(w, src + c) : wdt == YOLO_BF16 ? ldw<bf16_t>(w, src + c) : ldw<f16_t>(w, src + c);
    }
    __syncthreads();
    for (int g = 0; g < ngrp; ++g) {
        const PackJob& j = jobs[lead + g];
        const int ntaps = j.ntaps, Kpad = j.Kpad;
        T* out = (T*)j.out;
        if (j.mode == 0) {                      // out[o][t*I + i]
            const int ng = (in + 7) >> 3, units = on * ntaps * ng;
            const bool vec = (I & 7) == 0;
            for (int u = threadIdx.x; u < units; u += 256) {
                const int ig = u % ng, r2 = u / ng, t = r2 % ntaps, o = r2 / ntaps;
                const float* sp = &tile[o][ig * 8 * kk + stap[g][t]];
                T* dp = out + (long)(o0 + o) * Kpad + t * I + i0 + ig * 8;
                if (vec) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = sp[e * kk];
                    store_pack<T, 8>(dp, v);
                } else {
                    for (int e = 0; e < 8 && ig * 8 + e < in; ++e) dp[e] = from_f<T>(sp[e * kk]);
                }
            }
        } else {                                // out[i][t*O + o]
            const int ng = (on + 7) >> 3, units = in * ntaps * ng;
            const bool vec = (O & 7) == 0;
            for (int u = threadIdx.x; u < units; u += 256) {
                const int og = u % ng, r2 = u / ng, t = r2 % ntaps, i = r2 / ntaps;
                const float* sp = &tile[og * 8][i * kk + stap[g][t]];
                T* dp = out + (long)(i0 + i) * Kpad + t * O + o0 + og * 8;
                if (vec) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = sp[e * (TB * 9 + 1)];
                    store_pack<T, 8>(dp, v);
                } else {
                    for (int e = 0; e < 8 && og * 8 + e < on; ++e) dp[e] = from_f<T>(sp[e * (TB * 9 + 1)]);
                }
            }
        }
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

// second pass over the host job table: groups the consecutive jobs of one parameter (forward form, then its data-gradient
// classes) and assigns each group its first tile; returns the total tile (= workgroup) count.  The packed buffers' pad
// columns (Kpad beyond taps * inner) are NOT written by yolo_pack_batched: the caller zeroes the buffers once.
long yolo_pack_jobs_finalize(void* jobs_host, int njobs) {
    PackJob* jobs = (PackJob*)jobs_host;
    long c = 0;
    int i = 0;
    while (i < njobs) {
        int n = 1;
        while (i + n < njobs && n < MAXGRP && jobs[i + n].w == jobs[i].w && jobs[i + n].O == jobs[i].O && jobs[i + n].I == jobs[i].I &&
               jobs[i + n].k == jobs[i].k && jobs[i + n].mode == 1)
            ++n;
        for (int g = 0; g < n; ++g) {
            jobs[i + g].cstart = c;
            jobs[i + g].lead = i;
            jobs[i + g].ngrp = n;
        }
        c += (long)((jobs[i].O + TB - 1) / TB) * ((jobs[i].I + TB - 1) / TB);
        i += n;
    }
    return c;
}

int yolo_pack_batched(const void* jobs_dev, int njobs, long nchunks, int out_dtype, hipStream_t st) {
    if (njobs <= 0 || nchunks <= 0) return YOLO_OK;
    YOLO_DISPATCH_T(out_dtype, hipLaunchKernelGGL((k_pack_tiles<T>), dim3((unsigned)nchunks), dim3(256), 0, st, (const PackJob*)jobs_dev, njobs));
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
