// Copy / cast MANY tensors in one launch: the gradient exchange's pack (every parameter gradient -> its slice of the
// flat communication buffer, fp32 -> bf16 when the exchange is compressed) and unpack (back, optionally scaled by
// 1 / world size for backends without an averaging all-reduce).  Replaces torch._foreach_copy_ / _flatten_dense_tensors
// on the replayed data-parallel step (reference: DistributedDataParallel's bucket copies, src/training/utils_train.py:190).
// A device job table lists (src, dst, numel, dtypes) per tensor; one workgroup owns one 4096-element chunk of one
// tensor, found by binary search over the jobs' first chunks -- the scheme of k_adamw (optim.hip).
#include "common.h"

struct CopyJob {
    const void* src;
    void* dst;
    long n;             // elements
    long cstart;        // first 4096-element chunk (= workgroup) of this job
    int src_dtype, dst_dtype;
};

namespace {

constexpr int CHUNK = 4096;

__device__ __forceinline__ float ld_any(const void* p, int dt, long i) {
    return dt == YOLO_F32 ? ((const float*)p)[i] : dt == YOLO_BF16 ? to_f<bf16_t>(((const bf16_t*)p)[i]) : to_f<f16_t>(((const f16_t*)p)[i]);
}
__device__ __forceinline__ void st_any(void* p, int dt, long i, float x) {
    if (dt == YOLO_F32) ((float*)p)[i] = x;
    else if (dt == YOLO_BF16) ((bf16_t*)p)[i] = from_f<bf16_t>(x);
    else ((f16_t*)p)[i] = from_f<f16_t>(x);
}

template <typename S, typename D>
__device__ __forceinline__ void copy_vec(const CopyJob& j, long base, float scale) {
    // 8 elements per thread and trip: 16-byte packets on the 16-bit side, two on the fp32 side
    for (long i = base + threadIdx.x * 8L; i < base + CHUNK && i + 8 <= j.n; i += 256 * 8) {
        float v[8];
        load_pack<S, 8>((const S*)j.src + i, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= scale;
        store_pack<D, 8>((D*)j.dst + i, v);
    }
    const long tail = j.n & ~7L;                              // the last < 8 elements of the tensor
    if (tail >= base && tail < base + CHUNK && threadIdx.x < j.n - tail)
        st_any(j.dst, j.dst_dtype, tail + threadIdx.x, ld_any(j.src, j.src_dtype, tail + threadIdx.x) * scale);
}

__global__ __launch_bounds__(256) void k_multi_copy(const CopyJob* __restrict__ jobs, int njobs, float scale) {
    __shared__ int sj;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {                                     // last job with cstart <= blockIdx.x
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].cstart <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sj = lo;
    }
    __syncthreads();
    const CopyJob j = jobs[sj];
    const long base = ((long)blockIdx.x - j.cstart) * CHUNK;
    const bool al = ((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 31) == 0;
    if (al && j.src_dtype == YOLO_F32 && j.dst_dtype == YOLO_BF16) return copy_vec<float, bf16_t>(j, base, scale);
    if (al && j.src_dtype == YOLO_BF16 && j.dst_dtype == YOLO_F32) return copy_vec<bf16_t, float>(j, base, scale);
    if (al && j.src_dtype == YOLO_F32 && j.dst_dtype == YOLO_F32) return copy_vec<float, float>(j, base, scale);
    if (al && j.src_dtype == YOLO_BF16 && j.dst_dtype == YOLO_BF16) return copy_vec<bf16_t, bf16_t>(j, base, scale);
    for (long i = base + threadIdx.x; i < base + CHUNK && i < j.n; i += 256)
        st_any(j.dst, j.dst_dtype, i, ld_any(j.src, j.src_dtype, i) * scale);
}

}  // namespace

extern "C" {

int yolo_copy_job_bytes(void) { return (int)sizeof(CopyJob); }

int yolo_copy_job_fill(void* jobs_host, int index, const void* src, int src_dtype, void* dst, int dst_dtype, long n) {
    if (n < 0) return YOLO_ERR_ARG;
    CopyJob& j = ((CopyJob*)jobs_host)[index];
    j.src = src; j.dst = dst; j.n = n; j.cstart = 0; j.src_dtype = src_dtype; j.dst_dtype = dst_dtype;
    return YOLO_OK;
}

// assign the chunk ranges; returns the number of workgroups
long yolo_copy_jobs_finalize(void* jobs_host, int njobs) {
    CopyJob* jobs = (CopyJob*)jobs_host;
    long c = 0;
    for (int i = 0; i < njobs; ++i) {
        jobs[i].cstart = c;
        long nch = (jobs[i].n + CHUNK - 1) / CHUNK;
        c += nch > 0 ? nch : 1;
    }
    return c;
}

int yolo_multi_copy(const void* jobs_dev, int njobs, long nchunks, float scale, hipStream_t st) {
    if (njobs <= 0 || nchunks <= 0) return YOLO_OK;
    hipLaunchKernelGGL(k_multi_copy, dim3((unsigned)nchunks), dim3(256), 0, st, (const CopyJob*)jobs_dev, njobs, scale);
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
