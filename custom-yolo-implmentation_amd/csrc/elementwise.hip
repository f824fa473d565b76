// HBM-bound NHWC kernels: layout changes, channel copies, BatchNorm (+SiLU, +residual) forward and
// backward, SPPF max-pool, nearest x2 upsample.  One 16-byte packet per lane wherever the channel
// count allows it (guide: Guideline 13), fp32 arithmetic, fp32 statistics.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int RS_ROWS = 2;        // pixel rows (independent 16-byte loads per tensor) in flight per thread
constexpr int RS_MAXBLK = 256 * 8;  // row-strided grids: at most 8 workgroups per CU

template <typename T> __host__ bool vec_ok(const void* p, long ld, int C) {
    constexpr int V = vec_of<T>::N;
    return (C % V == 0) && (ld % V == 0) && ((reinterpret_cast<uintptr_t>(p) % 16) == 0);
}

// ---------------------------------------------------------------------------------------------
// (n, c, m) <-> NHWC tiled transposes.  "ncm" side: elem(n,c,p) at base[n*sn + c*sc + off + p].
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void k_ncm_to_nhwc(const TI* __restrict__ src, long sn, long sc, long off,
                              TO* __restrict__ dst, int ld, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows of 32
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? to_f<TI>(src[n * sn + c * sc + off + p]) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) dst[((long)n * HW + p) * ld + c] = from_f<TO>(tile[tx][r]);
    }
}

template <typename TI, typename TO>
__global__ void k_nhwc_to_ncm(const TI* __restrict__ src, int ld, TO* __restrict__ dst, long sn,
                              long sc, long off, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? to_f<TI>(src[((long)n * HW + p) * ld + c]) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) dst[n * sn + c * sc + off + p] = from_f<TO>(tile[tx][r]);
    }
}

// The head's six branch outputs <-> the (N, no, M) prediction tensor in ONE launch each way (HeadPack): a problem table in
// the kernel arguments, a workgroup finds its branch from the prefix of tile counts and transposes one 32 x 32 tile.
constexpr int HEAD_GROUP = 8;
struct HeadGroup {
    void* t[HEAD_GROUP];            // NHWC branch tensors
    int ld[HEAD_GROUP], C[HEAD_GROUP], HW[HEAD_GROUP], c_off[HEAD_GROUP], m_off[HEAD_GROUP], tp[HEAD_GROUP];   // tp: pixel tiles
    int start[HEAD_GROUP + 1];      // first workgroup (per image) of branch i
    int n;
};

template <typename T, bool PACK>
__global__ __launch_bounds__(256) void k_head_group(HeadGroup g, T* __restrict__ preds, long sn, long sc) {
    __shared__ float tile[32][33];
    int bi = 0;
#pragma unroll
    for (int i = 1; i < HEAD_GROUP; ++i)
        if (i < g.n && (int)blockIdx.x >= g.start[i]) bi = i;
    T* x; int ld, C, HW, c_off, m_off, tp, st;
#define HG_PICK(I) case I: x = (T*)g.t[I]; ld = g.ld[I]; C = g.C[I]; HW = g.HW[I]; c_off = g.c_off[I]; m_off = g.m_off[I]; tp = g.tp[I]; st = g.start[I]; break;
    switch (bi) {
        HG_PICK(1) HG_PICK(2) HG_PICK(3) HG_PICK(4) HG_PICK(5) HG_PICK(6) HG_PICK(7)
        default: x = (T*)g.t[0]; ld = g.ld[0]; C = g.C[0]; HW = g.HW[0]; c_off = g.c_off[0]; m_off = g.m_off[0]; tp = g.tp[0]; st = g.start[0]; break;
    }
#undef HG_PICK
    const int tl = (int)blockIdx.x - st;
    const int n = blockIdx.y, p0 = (tl % tp) * 32, c0 = (tl / tp) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    T* pr = preds + n * sn + (long)c_off * sc + m_off;
    if (PACK) {
        for (int r = ty; r < 32; r += 8) {
            const int p = p0 + r, c = c0 + tx;
            tile[r][tx] = (p < HW && c < C) ? to_f<T>(x[((long)n * HW + p) * ld + c]) : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, p = p0 + tx;
            if (c < C && p < HW) pr[c * sc + p] = from_f<T>(tile[tx][r]);
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, p = p0 + tx;
            tile[r][tx] = (c < C && p < HW) ? to_f<T>(pr[c * sc + p]) : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int p = p0 + r, c = c0 + tx;
            if (p < HW && c < C) x[((long)n * HW + p) * ld + c] = from_f<T>(tile[tx][r]);
        }
    }
}

// The same with 16-byte accesses on both sides (16-bit tensors, C % 8 == 0, HW / M / offsets multiples of 8): a workgroup
// owns 64 pixels x all C channels of one branch and image; the tile crosses LDS as [pixel][channel] (row stride C + 2
// elements: the 8 strided 2-byte reads / writes of a transposing thread fall in different banks).
template <typename T, bool PACK>
__global__ __launch_bounds__(256) void k_head_group_vec(HeadGroup g, T* __restrict__ preds, long sn, long sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hg_raw[];
    T* tile = reinterpret_cast<T*>(hg_raw);                   // [64][C + 2]
    int bi = 0;
#pragma unroll
    for (int i = 1; i < HEAD_GROUP; ++i)
        if (i < g.n && (int)blockIdx.x >= g.start[i]) bi = i;
    T* x; int ld, C, HW, c_off, m_off, st;
#define HG_PICK(I) case I: x = (T*)g.t[I]; ld = g.ld[I]; C = g.C[I]; HW = g.HW[I]; c_off = g.c_off[I]; m_off = g.m_off[I]; st = g.start[I]; break;
    switch (bi) {
        HG_PICK(1) HG_PICK(2) HG_PICK(3) HG_PICK(4) HG_PICK(5) HG_PICK(6) HG_PICK(7)
        default: x = (T*)g.t[0]; ld = g.ld[0]; C = g.C[0]; HW = g.HW[0]; c_off = g.c_off[0]; m_off = g.m_off[0]; st = g.start[0]; break;
    }
#undef HG_PICK
    const int n = blockIdx.y, p0 = ((int)blockIdx.x - st) * 64;
    const int LDT = C + 2, cpr = C / 8;                       // 16-byte chunks per pixel row
    T* pr = preds + n * sn + (long)c_off * sc + m_off;
    T* xr = x + ((long)n * HW + p0) * ld;
    const int npx = HW - p0 < 64 ? HW - p0 : 64;              // multiple of 8
    if (PACK) {
        for (int i = threadIdx.x; i < npx * cpr; i += 256) {
            const int p = i / cpr, ch = (i - p * cpr) * 8;
            const pack_t<T, 8> v = load_raw<T, 8>(xr + (long)p * ld + ch);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[p * LDT + ch + e] = v.v[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < C * (npx / 8); i += 256) {
            const int c = i / (npx / 8), pg = (i - c * (npx / 8)) * 8;
            pack_t<T, 8> v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] = tile[(pg + e) * LDT + c];
            *reinterpret_cast<pack_t<T, 8>*>(pr + c * sc + p0 + pg) = v;
        }
    } else {
        for (int i = threadIdx.x; i < C * (npx / 8); i += 256) {
            const int c = i / (npx / 8), pg = (i - c * (npx / 8)) * 8;
            const pack_t<T, 8> v = load_raw<T, 8>(pr + c * sc + p0 + pg);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[(pg + e) * LDT + c] = v.v[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < npx * cpr; i += 256) {
            const int p = i / cpr, ch = (i - p * cpr) * 8;
            pack_t<T, 8> v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] = tile[p * LDT + ch + e];
            *reinterpret_cast<pack_t<T, 8>*>(xr + (long)p * ld + ch) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// channel-slice copy / accumulate (concat, chunk backward, gradient fan-in)
// ---------------------------------------------------------------------------------------------
template <typename T, int V, bool ACC>
__global__ void k_copy_channels(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd,
                                long npix, int cv) {
    long total = npix * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / cv;
        int cg = (int)(i - p * cv);
        float a[V];
        load_pack<T, V>(src + p * lds_ + cg * V, a);
        if (ACC) {
            float b[V];
            load_pack<T, V>(dst + p * ldd + cg * V, b);
#pragma unroll
            for (int j = 0; j < V; ++j) a[j] += b[j];
        }
        store_pack<T, V>(dst + p * ldd + cg * V, a);
    }
}

// dst = sum of up to four channel-slice tensors (fp32 accumulate, one rounding): the gradient of a tensor with several
// consumers in ONE pass (autograd's own accumulation is one ATen add -- three passes over memory -- per extra consumer)
struct AddSrcs { const void* p[4]; int ld[4]; };
template <typename T, int V, int NS>
__global__ void k_add_n(AddSrcs s, T* __restrict__ dst, int ldd, long npix, int cv) {
    long total = npix * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / cv;
        int cg = (int)(i - p * cv);
        float a[V];
        load_pack<T, V>((const T*)s.p[0] + p * s.ld[0] + cg * V, a);
#pragma unroll
        for (int k = 1; k < NS; ++k) {
            float b[V];
            load_pack<T, V>((const T*)s.p[k] + p * s.ld[k] + cg * V, b);
#pragma unroll
            for (int j = 0; j < V; ++j) a[j] += b[j];
        }
        store_pack<T, V>(dst + p * ldd + cg * V, a);
    }
}

// ---------------------------------------------------------------------------------------------
// per-channel reductions over pixels: partial[blk][2][C]
//   MODE 0: (sum y, sum y^2)                      -- BN batch statistics / bias gradient
//   MODE 1: (sum dz, sum dz*y), dz = dout*act'(z), z = y*scale+shift   (the finalize kernel centres it)
// ---------------------------------------------------------------------------------------------
// sigmoid for the BatchNorm + SiLU kernels: hardware reciprocal (1 ulp) instead of the IEEE division -- the division
// expands to ~10 VALU instructions per element (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) and these kernels
// are co-limited by VALU issue (a wave64 op takes 4 cycles on the 16-lane SIMD): about a quarter of their instructions
__device__ __forceinline__ float sigmoid_rcp(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == 0) return 1.f;
    float s = sigmoid_rcp(z);
    return s * (1.f + z * (1.f - s));
}
__device__ __forceinline__ float act_fwd(float z, int act) { return act == 0 ? z : z * sigmoid_rcp(z); }

template <typename T, int V, int MODE>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_channel_reduce(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int ldd,
                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                 long npix, int C, int act, float* __restrict__ partial) {
    __shared__ float red[TPB][2 * V + 1];
    const int cv = C / V;
    const int tpr = cv < TPB ? cv : TPB;
    const int rpb = TPB / tpr;
    const int r = threadIdx.x / tpr;
    const int cg = blockIdx.y * tpr + (threadIdx.x - r * tpr);
    const bool active = (r < rpb) && (cg < cv);
    float s[V], q[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = q[j] = 0.f;
    if (active) {
        float sc[V], sh[V];
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) { sc[j] = scale[cg * V + j]; sh[j] = shift[cg * V + j]; }
        }
        const long step = (long)gridDim.x * rpb;
        auto one = [&](const pack_t<T, V>& pa, const pack_t<T, V>& pd) {
            float a[V], d[V];
            unpack<T, V>(pa, a);
            if (MODE == 1) unpack<T, V>(pd, d);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (MODE == 0) { s[j] += a[j]; q[j] += a[j] * a[j]; }
                else {
                    float dz = d[j] * act_grad(a[j] * sc[j] + sh[j], act);
                    s[j] += dz;
                    q[j] += dz * a[j];           // raw; k_bn_bwd_finalize turns it into sum(dz*yhat) in double
                }
            }
        };
        long p = (long)blockIdx.x * rpb + r;
        for (; p + (RS_ROWS - 1) * step < npix; p += RS_ROWS * step) {
            pack_t<T, V> ra[RS_ROWS], rd[RS_ROWS];
#pragma unroll
            for (int k = 0; k < RS_ROWS; ++k) {
                ra[k] = load_raw<T, V>(y + (p + k * step) * ldy + cg * V);
                if (MODE == 1) rd[k] = load_raw<T, V>(dout + (p + k * step) * ldd + cg * V);
            }
#pragma unroll
            for (int k = 0; k < RS_ROWS; ++k) one(ra[k], rd[k]);
        }
        for (; p < npix; p += step) {
            pack_t<T, V> pa = load_raw<T, V>(y + p * ldy + cg * V), pd;
            if (MODE == 1) pd = load_raw<T, V>(dout + p * ldd + cg * V);
            one(pa, pd);
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red[threadIdx.x][j] = s[j]; red[threadIdx.x][V + j] = q[j]; }
    __syncthreads();
    // one thread per (channel group, value): tpr*2V outputs, each the sum over the workgroup's rpb pixel rows
    // (a single row of threads walking all rows serially cost 10-20 us per workgroup on narrow layers)
    float* o = partial + (long)blockIdx.x * 2 * C;
    for (int t = threadIdx.x; t < tpr * 2 * V; t += TPB) {
        const int cl = t / (2 * V), j = t - cl * 2 * V;
        float a = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < rpb; ++rr) a += red[rr * tpr + cl][j];
        const int cgo = blockIdx.y * tpr + cl;
        if (cgo < cv) o[(j / V) * C + cgo * V + (j % V)] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// Training-mode BN without finalize launches.  Per-channel sums live in a caller-zeroed accumulator
// acc[BN_REPL][2][C] (fp32): producers (the conv epilogue, or k_channel_acc below) add their partial
// sums into replica (workgroup index mod BN_REPL) with float atomics -- spreading a layer's adds over
// 8 rows per channel keeps the global-atomic unit out of its contended regime -- and every consumer
// workgroup folds the replicas and derives the per-channel constants for its own channel group in
// its prologue.  Workgroup (0, y) also publishes mean / invstd / running statistics (forward) or
// dgamma / dbeta (backward).
// ---------------------------------------------------------------------------------------------
// BatchNorm parameters (gamma, beta -> dgamma, dbeta) and buffers (running statistics) in their own element type:
// fp32 under DDP / autocast, the low-precision parameter dtype under FSDP mixed precision, whose policy casts
// parameters AND buffers (reference src/training/utils_train.py:84-89,146-153).  Read / written once per channel in
// the prologues, arithmetic in fp32 / double as before, one rounding on the way out.
struct PIn {
    const void* p; int dt;
    __device__ __forceinline__ float ld(int i) const {
        return dt == YOLO_F32 ? ((const float*)p)[i] : dt == YOLO_BF16 ? (float)((const bf16_t*)p)[i] : (float)((const f16_t*)p)[i];
    }
};
struct PIo {
    void* p; int dt;
    __device__ __forceinline__ float ld(int i) const {
        return dt == YOLO_F32 ? ((const float*)p)[i] : dt == YOLO_BF16 ? (float)((const bf16_t*)p)[i] : (float)((const f16_t*)p)[i];
    }
    __device__ __forceinline__ void st(int i, float v) const {
        if (dt == YOLO_F32) ((float*)p)[i] = v; else if (dt == YOLO_BF16) ((bf16_t*)p)[i] = (bf16_t)v; else ((f16_t*)p)[i] = (f16_t)v;
    }
};
static inline bool pdt_ok(int dt) { return dt == YOLO_F32 || dt == YOLO_BF16 || dt == YOLO_F16; }

constexpr int BN_REPL = 8;

template <int V>
__device__ __forceinline__ void fold_replicas(const float* __restrict__ acc, int C, int c0, float (&s)[V], float (&q)[V]) {
    // issue every replica's loads before the first add: 16 independent packets per thread instead of a
    // chain of dependent scalar loads (which cost ~14 us per launch)
    float ts[BN_REPL][V], tq[BN_REPL][V];
#pragma unroll
    for (int r = 0; r < BN_REPL; ++r) {
        const float* a = acc + (long)r * 2 * C + c0;
        load_pack<float, V>(a, ts[r]);
        load_pack<float, V>(a + C, tq[r]);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        s[j] = ((ts[0][j] + ts[1][j]) + (ts[2][j] + ts[3][j])) + ((ts[4][j] + ts[5][j]) + (ts[6][j] + ts[7][j]));
        q[j] = ((tq[0][j] + tq[1][j]) + (tq[2][j] + tq[3][j])) + ((tq[4][j] + tq[5][j]) + (tq[6][j] + tq[7][j]));
    }
}

// MODE 0: acc += (sum y, sum y^2).  MODE 1: acc += (sum dz, sum dz*yhat) with z = (y-mean)*invstd*gamma + beta
template <typename T, int V, int MODE, int ACT>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_channel_acc(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int ldd,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              const float* __restrict__ mean, const float* __restrict__ invstd,
                              long npix, int C, int act, float* __restrict__ acc, int tpr, int rev) {
    const long pix_b = rev ? npix - 1 : 0, pix_d = rev ? -1 : 1;   // traversal direction (see bn_rev())
#define PIX(p_) (pix_b + pix_d * (p_))

    __shared__ float red[TPB][2 * V + 1];
    const int cv = C / V;
    const int rpb = TPB / tpr;
    const int r = threadIdx.x / tpr;
    const int cg = blockIdx.y * tpr + (threadIdx.x - r * tpr);
    const bool active = (r < rpb) && (cg < cv);
    float s[V], q[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = q[j] = 0.f;
    if (active) {
        float sc[V], sh[V];
        if (MODE == 1) {                                    // MODE 1: gamma / beta carry the forward's scale / shift
#pragma unroll
            for (int j = 0; j < V; ++j) { sc[j] = gamma[cg * V + j]; sh[j] = beta[cg * V + j]; }
        }
        const long step = (long)gridDim.x * rpb;
        auto one = [&](const pack_t<T, V>& pa, const pack_t<T, V>& pd) {
            float a[V], d[V];
            unpack<T, V>(pa, a);
            if (MODE == 1) unpack<T, V>(pd, d);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (MODE == 0) { s[j] += a[j]; q[j] += a[j] * a[j]; }
                else {
                    float dz = d[j] * act_grad(a[j] * sc[j] + sh[j], ACT);
                    s[j] += dz;
                    q[j] += dz * a[j];            // raw; the apply kernel centres it
                }
            }
        };
        long p = (long)blockIdx.x * rpb + r;
        for (; p + (RS_ROWS - 1) * step < npix; p += RS_ROWS * step) {
            pack_t<T, V> ra[RS_ROWS], rd[RS_ROWS];
#pragma unroll
            for (int k = 0; k < RS_ROWS; ++k) {
                ra[k] = load_raw<T, V>(y + PIX(p + k * step) * ldy + cg * V);
                if (MODE == 1) rd[k] = load_raw<T, V>(dout + PIX(p + k * step) * ldd + cg * V);
            }
#pragma unroll
            for (int k = 0; k < RS_ROWS; ++k) one(ra[k], rd[k]);
        }
        for (; p < npix; p += step) {
            pack_t<T, V> pa = load_raw<T, V>(y + PIX(p) * ldy + cg * V), pd;
            if (MODE == 1) pd = load_raw<T, V>(dout + PIX(p) * ldd + cg * V);
            one(pa, pd);
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red[threadIdx.x][j] = s[j]; red[threadIdx.x][V + j] = q[j]; }
    __syncthreads();
    float* o = acc + (long)(blockIdx.x % BN_REPL) * 2 * C;
    for (int t = threadIdx.x; t < tpr * 2 * V; t += TPB) {
        const int cl = t / (2 * V), j = t - cl * 2 * V;
        float a = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < rpb; ++rr) a += red[rr * tpr + cl][j];
        const int cgo = blockIdx.y * tpr + cl;
        if (cgo < cv) atomicAdd(o + (j / V) * C + cgo * V + (j % V), a);
    }
}
#undef PIX

// acc[8][2][C] -> mean, invstd, scale, shift (+ running statistics): one thread per channel, 16 loads
__global__ void k_bn_finalize_acc(const float* __restrict__ acc, float count, int C, const void* __restrict__ gamma_,
                                  const void* __restrict__ beta_, void* __restrict__ rmean_, void* __restrict__ rvar_,
                                  float momentum, float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                  float* __restrict__ scale, float* __restrict__ shift, int pdt, int bdt) {
    const PIn gamma{gamma_, pdt}; const PIn beta{beta_, pdt}; const PIo rmean{rmean_, bdt}, rvar{rvar_, bdt};
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int r = 0; r < BN_REPL; ++r) { s += acc[(long)r * 2 * C + c]; q += acc[(long)r * 2 * C + C + c]; }
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    const float g = gamma.ld(c) * is;
    scale[c] = g;
    shift[c] = beta.ld(c) - (float)m * g;
    if (rmean.p) {
        const double unb = count > 1.f ? var * (count / (count - 1.0)) : var;
        rmean.st(c, (1.f - momentum) * rmean.ld(c) + momentum * (float)m);
        rvar.st(c, (1.f - momentum) * rvar.ld(c) + momentum * (float)unb);
    }
}

// out = act(BN_batch(y)) (+ res) with the finalize folded in: every workgroup derives scale / shift of ITS channels
// from the statistics accumulator acc[8][2][C] (coalesced loads spread over the workgroup, then LDS); workgroup row 0
// also publishes mean / invstd / scale / shift and updates the running statistics.  Same arithmetic as
// k_bn_finalize_acc.
template <typename T, int V, int ACT>
__global__ __launch_bounds__(TPB) void k_bn_act_fwd_train(const T* __restrict__ y, int ldy, const float* __restrict__ acc, float count,
                        const void* __restrict__ gamma_, const void* __restrict__ beta_, void* __restrict__ rmean_,
                        void* __restrict__ rvar_, float momentum, float eps, float* __restrict__ mean_out,
                        float* __restrict__ invstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out,
                        const T* __restrict__ res, int ldr, T* __restrict__ out, int ldo, long npix, int C, int act, int tpr, int pdt, int bdt, int rev) {
    const long pix_b = rev ? npix - 1 : 0, pix_d = rev ? -1 : 1;   // traversal direction (see bn_rev())
#define PIX(p_) (pix_b + pix_d * (p_))

    const PIn gamma{gamma_, pdt}; const PIn beta{beta_, pdt}; const PIo rmean{rmean_, bdt}, rvar{rvar_, bdt};
    extern __shared__ float cf[];                            // [2][cw]: scale, shift of this workgroup's channels
    const int cv = C / V;
    const int rpb = TPB / tpr;
    const int cw = tpr * V;
    for (int t = threadIdx.x; t < cw; t += TPB) {
        const int c = blockIdx.y * cw + t;
        float g = 0.f, sh0 = 0.f;
        if (c < C) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int r = 0; r < BN_REPL; ++r) { s += acc[(long)r * 2 * C + c]; q += acc[(long)r * 2 * C + C + c]; }
            const double m = s / count;
            double var = q / count - m * m;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)eps));
            g = gamma.ld(c) * is;
            sh0 = beta.ld(c) - (float)m * g;
            if (blockIdx.x == 0) {
                mean_out[c] = (float)m;
                invstd_out[c] = is;
                scale_out[c] = g;
                shift_out[c] = sh0;
                if (rmean.p) {
                    const double unb = count > 1.f ? var * (count / (count - 1.0)) : var;
                    rmean.st(c, (1.f - momentum) * rmean.ld(c) + momentum * (float)m);
                    rvar.st(c, (1.f - momentum) * rvar.ld(c) + momentum * (float)unb);
                }
            }
        }
        cf[t] = g;
        cf[cw + t] = sh0;
    }
    __syncthreads();
    const int r = threadIdx.x / tpr;
    const int cl = threadIdx.x - r * tpr;
    const int cg = blockIdx.y * tpr + cl;
    if (r >= rpb || cg >= cv) return;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = cf[cl * V + j]; sh[j] = cf[cw + cl * V + j]; }
    const long step = (long)gridDim.x * rpb;
    auto one = [&](long p, const pack_t<T, V>& pa, const pack_t<T, V>& pt) {
        float a[V], t[V];
        unpack<T, V>(pa, a);
        if (res) unpack<T, V>(pt, t);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] = act_fwd(a[j] * sc[j] + sh[j], ACT) + (res ? t[j] : 0.f);
        store_pack<T, V>(out + PIX(p) * ldo + cg * V, a);
    };
    long p = (long)blockIdx.x * rpb + r;
    for (; p + (RS_ROWS - 1) * step < npix; p += RS_ROWS * step) {
        pack_t<T, V> ra[RS_ROWS], rt[RS_ROWS];
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) {
            ra[k] = load_raw<T, V>(y + PIX(p + k * step) * ldy + cg * V);
            if (res) rt[k] = load_raw<T, V>(res + PIX(p + k * step) * ldr + cg * V);
        }
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) one(p + k * step, ra[k], rt[k]);
    }
    for (; p < npix; p += step) {
        pack_t<T, V> pa = load_raw<T, V>(y + PIX(p) * ldy + cg * V), pt;
        if (res) pt = load_raw<T, V>(res + PIX(p) * ldr + cg * V);
        one(p, pa, pt);
    }
}
#undef PIX

// dy = A*dz + B*y + D with the backward finalize folded in: acc[8][2][C] holds (sum dz, sum dz*y) of the reduction
// kernel (float atomics into 8 replicas); every workgroup folds the replicas of ITS channels, centres the second
// sum (sum dz*yhat = invstd*(sum dz*y - mean*sum dz)) and derives the five constants into LDS; workgroup row 0
// also writes dgamma / dbeta.  Same arithmetic as k_bn_bwd_finalize (double).
template <typename T, int V, int ACT>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(5, 8)))
void k_bn_act_bwd_apply_train(const T* __restrict__ dout, int ldd, const T* __restrict__ y, int ldy,
                              const float* __restrict__ scale, const float* __restrict__ shift,
                              const void* __restrict__ gamma_, const float* __restrict__ mean,
                              const float* __restrict__ invstd, const float* __restrict__ acc, float count,
                              void* __restrict__ dgamma_, void* __restrict__ dbeta_, T* __restrict__ dy, int lddy,
                              long npix, int C, int act, int tpr, int pdt, int rev) {
    const long pix_b = rev ? npix - 1 : 0, pix_d = rev ? -1 : 1;   // traversal direction (see bn_rev())
#define PIX(p_) (pix_b + pix_d * (p_))

    const PIn gamma{gamma_, pdt}; const PIo dgamma{dgamma_, pdt}, dbeta{dbeta_, pdt};
    extern __shared__ float cf[];                            // [5][cw]: scale, shift, A, B, D
    const int cv = C / V;
    const int rpb = TPB / tpr;
    const int cw = tpr * V;
    for (int t = threadIdx.x; t < cw; t += TPB) {
        const int c = blockIdx.y * cw + t;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (c < C) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int r = 0; r < BN_REPL; ++r) { s += acc[(long)r * 2 * C + c]; q += acc[(long)r * 2 * C + C + c]; }
            const double is = invstd[c], mu = mean[c];
            q = is * (q - mu * s);                           // sum(dz*yhat)
            const double k0 = (double)gamma.ld(c) * is, c1 = s / count, c2 = q / count;
            v0 = scale[c]; v1 = shift[c];
            v2 = (float)k0;
            v3 = (float)(-k0 * c2 * is);
            v4 = (float)(-k0 * c1 + k0 * c2 * mu * is);
            if (blockIdx.x == 0) { dbeta.st(c, (float)s); dgamma.st(c, (float)q); }
        }
        cf[t] = v0; cf[cw + t] = v1; cf[2 * cw + t] = v2; cf[3 * cw + t] = v3; cf[4 * cw + t] = v4;
    }
    __syncthreads();
    const int r = threadIdx.x / tpr;
    const int cl = threadIdx.x - r * tpr;
    const int cg = blockIdx.y * tpr + cl;
    if (r >= rpb || cg >= cv) return;
    const float* my = cf + cl * V;
    const long step = (long)gridDim.x * rpb;
    auto one = [&](long p, const pack_t<T, V>& pa, const pack_t<T, V>& pd) {
        float a[V], d[V];
        unpack<T, V>(pa, a);
        unpack<T, V>(pd, d);
#pragma unroll
        for (int j = 0; j < V; ++j)
            d[j] = my[2 * cw + j] * (d[j] * act_grad(a[j] * my[j] + my[cw + j], ACT)) + my[3 * cw + j] * a[j] + my[4 * cw + j];
        store_pack<T, V>(dy + PIX(p) * lddy + cg * V, d);
    };
    long p = (long)blockIdx.x * rpb + r;
    for (; p + (RS_ROWS - 1) * step < npix; p += RS_ROWS * step) {
        pack_t<T, V> ra[RS_ROWS], rd[RS_ROWS];
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) {
            ra[k] = load_raw<T, V>(y + PIX(p + k * step) * ldy + cg * V);
            rd[k] = load_raw<T, V>(dout + PIX(p + k * step) * ldd + cg * V);
        }
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) one(p + k * step, ra[k], rd[k]);
    }
    for (; p < npix; p += step)
        one(p, load_raw<T, V>(y + PIX(p) * ldy + cg * V), load_raw<T, V>(dout + PIX(p) * ldd + cg * V));
}
#undef PIX

// Finalize kernels run as (32 channels x 32 parts) 1024-thread workgroups: part j sums partial blocks
// j, j+32, ... of its channel (128-byte coalesced rows), LDS combines the 32 parts; the thread with
// part 0 gets the totals and returns its channel, the others return -1.  (A one-thread-per-channel
// loop over 512 partials cost ~95 us per launch, 15 ms per training step.)
constexpr int FIN_CH = 16, FIN_PARTS = 64;
__device__ __forceinline__ int fin_reduce(const float* __restrict__ partial, int nblk, int C, double& s, double& q) {
    __shared__ double red[2][FIN_PARTS][FIN_CH + 1];
    const int cl = threadIdx.x & (FIN_CH - 1), part = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    double ls = 0.0, lq = 0.0;
    if (c < C)
#pragma unroll 8
        for (int b = part; b < nblk; b += FIN_PARTS) {
            ls += partial[(long)b * 2 * C + c];
            lq += partial[(long)b * 2 * C + C + c];
        }
    red[0][part][cl] = ls;
    red[1][part][cl] = lq;
    __syncthreads();
    if (part != 0 || c >= C) return -1;
    s = 0.0; q = 0.0;
    for (int j = 0; j < FIN_PARTS; ++j) { s += red[0][j][cl]; q += red[1][j][cl]; }
    return c;
}

// finalize for training-mode BN: batch mean / biased var -> invstd, scale, shift; running stats
// updated with momentum and the unbiased variance (torch.nn.BatchNorm2d semantics).
__global__ void k_bn_finalize(const float* __restrict__ partial, int nblk, float count, int C,
                              const void* __restrict__ gamma_, const void* __restrict__ beta_,
                              void* __restrict__ rmean_, void* __restrict__ rvar_, float momentum, float eps,
                              float* __restrict__ mean, float* __restrict__ invstd,
                              float* __restrict__ scale, float* __restrict__ shift, int pdt, int bdt) {
    const PIn gamma{gamma_, pdt}; const PIn beta{beta_, pdt}; const PIo rmean{rmean_, bdt}, rvar{rvar_, bdt};
    double s, q;
    const int c = fin_reduce(partial, nblk, C, s, q);
    if (c < 0) return;
    double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    float g = gamma.ld(c) * is;
    scale[c] = g;
    shift[c] = beta.ld(c) - (float)m * g;
    if (rmean.p) {
        double unb = count > 1.f ? var * (count / (count - 1.0)) : var;
        rmean.st(c, (1.f - momentum) * rmean.ld(c) + momentum * (float)m);
        rvar.st(c, (1.f - momentum) * rvar.ld(c) + momentum * (float)unb);
    }
}

__global__ void k_bn_eval_coeffs(const void* __restrict__ gamma_, const void* __restrict__ beta_,
                                 const void* __restrict__ rmean_, const void* __restrict__ rvar_, float eps,
                                 int C, float* __restrict__ scale, float* __restrict__ shift, int pdt, int bdt) {
    const PIn gamma{gamma_, pdt}; const PIn beta{beta_, pdt}; const PIn rmean{rmean_, bdt}, rvar{rvar_, bdt};
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float g = gamma.ld(c) / sqrtf(rvar.ld(c) + eps);
    scale[c] = g;
    shift[c] = beta.ld(c) - rmean.ld(c) * g;
}

// backward finalize: dgamma = sum dz*yhat, dbeta = sum dz, coef = [gamma*invstd, dbeta/m, dgamma/m]
__global__ void k_bn_bwd_finalize(const float* __restrict__ partial, int nblk, float count, int C,
                                  const void* __restrict__ gamma_, const float* __restrict__ mean, const float* __restrict__ invstd,
                                  void* __restrict__ dgamma_, void* __restrict__ dbeta_, float* __restrict__ coef, int pdt) {
    const PIn gamma{gamma_, pdt}; const PIo dgamma{dgamma_, pdt}, dbeta{dbeta_, pdt};
    double s, q;
    const int c = fin_reduce(partial, nblk, C, s, q);
    if (c < 0) return;
    q = (double)invstd[c] * (q - (double)mean[c] * s);      // partial rows hold sum(dz*y): -> sum(dz*yhat)
    dbeta.st(c, (float)s);
    dgamma.st(c, (float)q);
    // dy = k0*(dz - c1 - yhat*c2), yhat = (y-mean)*invstd  ==  A*dz + B*y + D  (three constants per channel)
    const double k0 = (double)gamma.ld(c) * invstd[c], c1 = s / count, c2 = q / count;
    coef[c] = (float)k0;
    coef[C + c] = (float)(-k0 * c2 * invstd[c]);
    coef[2 * C + c] = (float)(-k0 * c1 + k0 * c2 * (double)mean[c] * invstd[c]);
}

__global__ void k_sum_finalize(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
    double s, q;
    const int c = fin_reduce(partial, nblk, C, s, q);
    if (c >= 0) out[c] = (float)s;
}

// out = act(y*scale + shift) (+ res)
template <typename T, int V>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bn_act_fwd(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                             const float* __restrict__ shift, const T* __restrict__ res, int ldr,
                             T* __restrict__ out, int ldo, long npix, int cv, int act) {
    // row-strided: a thread keeps ONE channel group (its scale/shift live in registers) and walks pixels;
    // no per-element index division, RS_ROWS pixels (loads) in flight per iteration
    const int tpr = cv < TPB ? cv : TPB, rpb = TPB / tpr;
    const int r = threadIdx.x / tpr;
    const int cg = blockIdx.y * tpr + (threadIdx.x - r * tpr);
    if (r >= rpb || cg >= cv) return;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = scale[cg * V + j]; sh[j] = shift[cg * V + j]; }
    const long step = (long)gridDim.x * rpb;
    auto one = [&](long p, const pack_t<T, V>& pa, const pack_t<T, V>& pt) {
        float a[V], t[V];
        unpack<T, V>(pa, a);
        if (res) unpack<T, V>(pt, t);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] = act_fwd(a[j] * sc[j] + sh[j], act) + (res ? t[j] : 0.f);
        store_pack<T, V>(out + p * ldo + cg * V, a);
    };
    long p = (long)blockIdx.x * rpb + r;
    for (; p + (RS_ROWS - 1) * step < npix; p += RS_ROWS * step) {       // RS_ROWS loads in flight, no bounds tests
        pack_t<T, V> ra[RS_ROWS], rt[RS_ROWS];
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) {
            ra[k] = load_raw<T, V>(y + (p + k * step) * ldy + cg * V);
            if (res) rt[k] = load_raw<T, V>(res + (p + k * step) * ldr + cg * V);
        }
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) one(p + k * step, ra[k], rt[k]);
    }
    for (; p < npix; p += step) {
        pack_t<T, V> pa = load_raw<T, V>(y + p * ldy + cg * V), pt;
        if (res) pt = load_raw<T, V>(res + p * ldr + cg * V);
        one(p, pa, pt);
    }
}

// dy = k0*(dz - c1 - yhat*c2);  eval-style (coef == null): dy = scale*dz
template <typename T, int V>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_bn_act_bwd_apply(const T* __restrict__ dout, int ldd, const T* __restrict__ y, int ldy,
                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                   const float* __restrict__ coef, T* __restrict__ dy, int lddy,
                                   long npix, int C, int act) {
    // dy = A*dz + B*y + D with (A, B, D) = coef rows from k_bn_bwd_finalize; frozen statistics: (scale, 0, 0).
    // The five per-channel constants of the workgroup's channels sit in LDS ([5][tpr*V] floats, dynamic) and are
    // read at use: forty live coefficient registers per thread would halve the occupancy of a pure streaming kernel.
    extern __shared__ float cf[];
    const int cv = C / V;
    const int tpr = cv < TPB ? cv : TPB, rpb = TPB / tpr;
    const int cw = tpr * V;                                  // channels of this workgroup
    for (int t = threadIdx.x; t < cw; t += TPB) {
        const int c = blockIdx.y * cw + t;
        const bool in = c < C;
        cf[t] = in ? scale[c] : 0.f;
        cf[cw + t] = in ? shift[c] : 0.f;
        cf[2 * cw + t] = in ? (coef ? coef[c] : scale[c]) : 0.f;
        cf[3 * cw + t] = (in && coef) ? coef[C + c] : 0.f;
        cf[4 * cw + t] = (in && coef) ? coef[2 * C + c] : 0.f;
    }
    __syncthreads();
    const int r = threadIdx.x / tpr;
    const int cl = threadIdx.x - r * tpr;
    const int cg = blockIdx.y * tpr + cl;
    if (r >= rpb || cg >= cv) return;
    const float* my = cf + cl * V;
    const long step = (long)gridDim.x * rpb;
    auto one = [&](long p, const pack_t<T, V>& pa, const pack_t<T, V>& pd) {
        float a[V], d[V];
        unpack<T, V>(pa, a);
        unpack<T, V>(pd, d);
#pragma unroll
        for (int j = 0; j < V; ++j)
            d[j] = my[2 * cw + j] * (d[j] * act_grad(a[j] * my[j] + my[cw + j], act)) + my[3 * cw + j] * a[j] + my[4 * cw + j];
        store_pack<T, V>(dy + p * lddy + cg * V, d);
    };
    long p = (long)blockIdx.x * rpb + r;
    for (; p + (RS_ROWS - 1) * step < npix; p += RS_ROWS * step) {
        pack_t<T, V> ra[RS_ROWS], rd[RS_ROWS];
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) {
            ra[k] = load_raw<T, V>(y + (p + k * step) * ldy + cg * V);
            rd[k] = load_raw<T, V>(dout + (p + k * step) * ldd + cg * V);
        }
#pragma unroll
        for (int k = 0; k < RS_ROWS; ++k) one(p + k * step, ra[k], rd[k]);
    }
    for (; p < npix; p += step)
        one(p, load_raw<T, V>(y + p * ldy + cg * V), load_raw<T, V>(dout + p * ldd + cg * V));
}

// ---------------------------------------------------------------------------------------------
// SPPF 5x5 stride-1 pad-2 max pool; idx = kh*5+kw of the FIRST maximum in window scan order
// (ATen max_pool2d: strictly-greater update, NaN propagates) so ties route gradient identically.
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void k_maxpool5_fwd(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo,
                               uint8_t* __restrict__ idx, int N, int H, int W, int C) {
    const int cv = C / V;
    long total = (long)N * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / cv;
        int cg = (int)(i - p * cv);
        int w = (int)(p % W);
        long t = p / W;
        int h = (int)(t % H);
        long n = t / H;
        float best[V];
        int bi[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { best[j] = -INFINITY; bi[j] = -1; }
        // a row of the window = five UNCONDITIONAL loads (clamped addresses) issued together, then the compare chain in
        // ATen's scan order on the taps that exist; the first form (a branch in front of each of the 25 loads) took 26 us
        // on the 6.5 MB SPPF maps
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
            const int hh = h + kh - 2;
            const bool rok = hh >= 0 && hh < H;
            const int hc = hh < 0 ? 0 : (hh >= H ? H - 1 : hh);
            pack_t<T, V> raw[5];
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const int ww = w + kw - 2;
                const int wc = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
                raw[kw] = load_raw<T, V>(x + ((n * H + hc) * W + wc) * ldx + cg * V);
            }
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const int ww = w + kw - 2;
                if (!(rok && ww >= 0 && ww < W)) continue;
                float a[V];
                unpack<T, V>(raw[kw], a);
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (a[j] > best[j] || a[j] != a[j] || bi[j] < 0) { best[j] = a[j]; bi[j] = kh * 5 + kw; }
            }
        }
        store_pack<T, V>(out + p * ldo + cg * V, best);
        pack_t<uint8_t, V> pi;                                // the V argmax bytes in one store
#pragma unroll
        for (int j = 0; j < V; ++j) pi.v[j] = (uint8_t)bi[j];
        *reinterpret_cast<pack_t<uint8_t, V>*>(idx + p * C + cg * V) = pi;
    }
}

// gather form of the backward: dx(p) = sum over the <=25 windows containing p whose argmax is p
template <typename T, int V, bool ACC>
__global__ void k_maxpool5_bwd(const T* __restrict__ dout, int ldd, const uint8_t* __restrict__ idx,
                               T* __restrict__ dx, int ldx, int N, int H, int W, int C) {
    const int cv = C / V;
    long total = (long)N * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / cv;
        int cg = (int)(i - p * cv);
        int w = (int)(p % W);
        long t = p / W;
        int h = (int)(t % H);
        long n = t / H;
        float g[V];
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
            const int oh = h - kh + 2;        // output row whose window tap kh lands on h
            const bool rok = oh >= 0 && oh < H;
            const int oc = oh < 0 ? 0 : (oh >= H ? H - 1 : oh);
            pack_t<T, V> raw[5];
            pack_t<uint8_t, V> ri[5];
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {      // five unconditional (clamped) gradient + index loads per row, then the sums
                const int ow = w - kw + 2;
                const int wc = ow < 0 ? 0 : (ow >= W ? W - 1 : ow);
                const long q = (n * H + oc) * W + wc;
                raw[kw] = load_raw<T, V>(dout + q * ldd + cg * V);
                ri[kw] = *reinterpret_cast<const pack_t<uint8_t, V>*>(idx + q * C + cg * V);
            }
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const int ow = w - kw + 2;
                if (!(rok && ow >= 0 && ow < W)) continue;
                float d[V];
                unpack<T, V>(raw[kw], d);
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (ri[kw].v[j] == kh * 5 + kw) g[j] += d[j];
            }
        }
        if (ACC) {
            float b[V];
            load_pack<T, V>(dx + p * ldx + cg * V, b);
#pragma unroll
            for (int j = 0; j < V; ++j) g[j] += b[j];
        }
        store_pack<T, V>(dx + p * ldx + cg * V, g);
    }
}

// nearest x2 upsample: out(2H,2W); backward sums the 2x2 block
template <typename T, int V>
__global__ void k_upsample2x_fwd(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo,
                                 int N, int H, int W, int C) {
    const int cv = C / V;
    const int OW = 2 * W, OH = 2 * H;
    long total = (long)N * OH * OW * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / cv;
        int cg = (int)(i - p * cv);
        int ow = (int)(p % OW);
        long t = p / OW;
        int oh = (int)(t % OH);
        long n = t / OH;
        pack_t<T, V> v = *reinterpret_cast<const pack_t<T, V>*>(x + ((n * H + (oh >> 1)) * W + (ow >> 1)) * ldx + cg * V);
        *reinterpret_cast<pack_t<T, V>*>(out + p * ldo + cg * V) = v;
    }
}

template <typename T, int V, bool ACC>
__global__ void k_upsample2x_bwd(const T* __restrict__ dout, int ldd, T* __restrict__ dx, int ldx,
                                 int N, int H, int W, int C) {
    const int cv = C / V;
    const int OW = 2 * W, OH = 2 * H;
    long total = (long)N * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / cv;
        int cg = (int)(i - p * cv);
        int w = (int)(p % W);
        long t = p / W;
        int h = (int)(t % H);
        long n = t / H;
        float g[V];
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float d[V];
                load_pack<T, V>(dout + ((n * OH + 2 * h + a) * OW + 2 * w + b) * ldd + cg * V, d);
#pragma unroll
                for (int j = 0; j < V; ++j) g[j] += d[j];
            }
        if (ACC) {
            float b2[V];
            load_pack<T, V>(dx + p * ldx + cg * V, b2);
#pragma unroll
            for (int j = 0; j < V; ++j) g[j] += b2[j];
        }
        store_pack<T, V>(dx + p * ldx + cg * V, g);
    }
}

// grid of a row-strided kernel: x walks pixel rows (RS_ROWS per thread per iteration), y covers channel groups >
// 256.  At most RS_MAXBLK workgroups (all resident at once: no partial last round), and every workgroup gets
// the same number of iterations.
inline dim3 rs_grid(long npix, int cv) {
    int tpr = cv < TPB ? cv : TPB, rpb = TPB / tpr;
    long units = (npix + (long)RS_ROWS * rpb - 1) / ((long)RS_ROWS * rpb);
    if (units < 1) units = 1;
    long iters = (units + RS_MAXBLK - 1) / RS_MAXBLK;
    long gx = (units + iters - 1) / iters;
    return dim3((unsigned)gx, (unsigned)ceil_div(cv, tpr));
}

// Plan of the training-path BatchNorm kernels (the accumulator forms above).  Every workgroup pays a prologue /
// epilogue per channel it covers (16 accumulator loads + double arithmetic, or an LDS reduction + atomics), whatever
// the tensor's size: on a 20x20 map with 512 channels a workgroup that spans all channels (64 lanes per row) and two
// rows per lane fetched 32 KB of accumulators for 8 KB of activations.  Wide layers on small maps therefore use
// 16-lane rows (256-byte segments, channel groups on blockIdx.y): measured 12.1 -> 10.2 us forward and 30.6 -> 24.6 us
// backward for 512 channels at 20x20 x 32 images; capping the workgroup count as well made the 64..128-channel
// layers slower and is not done.
// Traversal direction of the three training-mode BatchNorm passes, one bit each (1 = from the last pixel down): bit 0 the
// forward normalise pass, bit 1 the backward reduction, bit 2 the backward apply.  A pass that walks a tensor in the
// direction OPPOSITE to the pass that last touched it meets that pass's most recent lines first -- still in the L2s / the
// 256 MB Infinity Cache -- instead of evicting them on its way to them.  YOLO_BN_REV overrides (A/B runs).
static inline int bn_rev() {
    static const int v = [] { const char* e = getenv("YOLO_BN_REV"); return e ? atoi(e) : 0; }();
    return v;
}
struct RsPlan { int tpr; dim3 grid; };
inline RsPlan rs_plan(long npix, int cv) {
    RsPlan p;
    p.tpr = cv < TPB ? cv : TPB;
    if (p.tpr > 16 && npix * cv <= (1L << 20)) p.tpr = 16;   // <= 8 M elements
    const int rpb = TPB / p.tpr, gy = ceil_div(cv, p.tpr);
    long units = (npix + (long)RS_ROWS * rpb - 1) / ((long)RS_ROWS * rpb);
    if (units < 1) units = 1;
    long gmax = RS_MAXBLK / gy;
    if (gmax < 8) gmax = 8;
    const long iters = (units + gmax - 1) / gmax;
    p.grid = dim3((unsigned)((units + iters - 1) / iters), (unsigned)gy);
    return p;
}

inline int ew_grid(long total) {
    long b = (total + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));   // <= 16 blocks per CU, grid-stride the rest
}

template <typename TI>
int ncm_to_nhwc_out(const void* src, long sn, long sc, long off, void* dst, int dst_dtype, int ld, int N,
                    int C, int HW, hipStream_t st) {
    dim3 g(ceil_div(HW, 32), ceil_div(C, 32), N);
    YOLO_DISPATCH_T(dst_dtype, hipLaunchKernelGGL((k_ncm_to_nhwc<TI, T>), g, dim3(256), 0, st, (const TI*)src,
                                                  sn, sc, off, (T*)dst, ld, C, HW));
    return YOLO_LAUNCH_CHECK();
}
template <typename TI>
int nhwc_to_ncm_out(const void* src, int ld, void* dst, int dst_dtype, long sn, long sc, long off, int N,
                    int C, int HW, hipStream_t st) {
    dim3 g(ceil_div(HW, 32), ceil_div(C, 32), N);
    YOLO_DISPATCH_T(dst_dtype, hipLaunchKernelGGL((k_nhwc_to_ncm<TI, T>), g, dim3(256), 0, st, (const TI*)src,
                                                  ld, (T*)dst, sn, sc, off, C, HW));
    return YOLO_LAUNCH_CHECK();
}

}  // namespace

#define PICK_V(T, ok, ...)                                       \
    if (ok) { constexpr int V = vec_of<T>::N; __VA_ARGS__; }     \
    else    { constexpr int V = 1; __VA_ARGS__; }

extern "C" {

int yolo_memset0(void* p, size_t bytes, hipStream_t st) { return yolo_zero_async(p, bytes, st); }

int yolo_ncm_to_nhwc(const void* src, int src_dtype, long sn, long sc, long off, void* dst, int dst_dtype,
                     int ld, int N, int C, int HW, hipStream_t st) {
    switch (src_dtype) {
        case YOLO_F32:  return ncm_to_nhwc_out<float>(src, sn, sc, off, dst, dst_dtype, ld, N, C, HW, st);
        case YOLO_BF16: return ncm_to_nhwc_out<bf16_t>(src, sn, sc, off, dst, dst_dtype, ld, N, C, HW, st);
        case YOLO_F16:  return ncm_to_nhwc_out<f16_t>(src, sn, sc, off, dst, dst_dtype, ld, N, C, HW, st);
    }
    return YOLO_ERR_DTYPE;
}

// up to 8 NHWC branch tensors (one dtype) <-> preds (N, cp, M) of the same dtype: branch i occupies channels
// [c_off[i], c_off[i] + C[i]) and anchors [m_off[i], m_off[i] + HW[i]).  pack != 0: branches -> preds; 0: preds -> branches.
int yolo_head_group(int pack, int n, void* const* branches, const int* lds, const int* Cs, const int* HWs, const int* c_offs,
                    const int* m_offs, void* preds, int cp, int M, int N, int dtype, hipStream_t st) {
    if (n < 1 || n > HEAD_GROUP) return YOLO_ERR_ARG;
    HeadGroup g;
    int wgs = 0;
    for (int i = 0; i < HEAD_GROUP; ++i) {
        const int j = i < n ? i : 0;
        g.t[i] = branches[j]; g.ld[i] = lds[j]; g.C[i] = Cs[j]; g.HW[i] = HWs[j]; g.c_off[i] = c_offs[j]; g.m_off[i] = m_offs[j];
        g.tp[i] = ceil_div(HWs[j], 32);
        g.start[i] = wgs;
        if (i < n) wgs += g.tp[i] * ceil_div(Cs[j], 32);
    }
    g.start[HEAD_GROUP] = wgs;
    g.n = n;
    if (wgs == 0 || N == 0) return YOLO_OK;
    // 16-bit tensors with 8-element alignment everywhere: the 16-byte form, 64 pixels x all channels per workgroup
    bool vec = dtype != YOLO_F32 && M % 8 == 0 && (reinterpret_cast<uintptr_t>(preds) & 15) == 0;
    int cmax = 0;
    for (int i = 0; i < n && vec; ++i) {
        vec = Cs[i] % 8 == 0 && HWs[i] % 8 == 0 && m_offs[i] % 8 == 0 && lds[i] % 8 == 0 && (reinterpret_cast<uintptr_t>(branches[i]) & 15) == 0;
        cmax = Cs[i] > cmax ? Cs[i] : cmax;
    }
    if (vec && (size_t)64 * (cmax + 2) * 2 <= 48 * 1024) {
        int w2 = 0;
        for (int i = 0; i < HEAD_GROUP; ++i) {
            const int j = i < n ? i : 0;
            g.start[i] = w2;
            if (i < n) w2 += ceil_div(HWs[j], 64);
        }
        g.start[HEAD_GROUP] = w2;
        const dim3 grid2((unsigned)w2, (unsigned)N);
        const size_t lds_bytes = (size_t)64 * (cmax + 2) * 2;
        if (dtype == YOLO_BF16) {
            if (pack) hipLaunchKernelGGL((k_head_group_vec<bf16_t, true>), grid2, dim3(256), lds_bytes, st, g, (bf16_t*)preds, (long)cp * M, (long)M);
            else hipLaunchKernelGGL((k_head_group_vec<bf16_t, false>), grid2, dim3(256), lds_bytes, st, g, (bf16_t*)preds, (long)cp * M, (long)M);
        } else {
            if (pack) hipLaunchKernelGGL((k_head_group_vec<f16_t, true>), grid2, dim3(256), lds_bytes, st, g, (f16_t*)preds, (long)cp * M, (long)M);
            else hipLaunchKernelGGL((k_head_group_vec<f16_t, false>), grid2, dim3(256), lds_bytes, st, g, (f16_t*)preds, (long)cp * M, (long)M);
        }
        return YOLO_LAUNCH_CHECK();
    }
    const dim3 grid((unsigned)wgs, (unsigned)N);
    YOLO_DISPATCH_T(dtype, {
        if (pack) hipLaunchKernelGGL((k_head_group<T, true>), grid, dim3(256), 0, st, g, (T*)preds, (long)cp * M, (long)M);
        else hipLaunchKernelGGL((k_head_group<T, false>), grid, dim3(256), 0, st, g, (T*)preds, (long)cp * M, (long)M);
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_nhwc_to_ncm(const void* src, int src_dtype, int ld, void* dst, int dst_dtype, long sn, long sc,
                     long off, int N, int C, int HW, hipStream_t st) {
    switch (src_dtype) {
        case YOLO_F32:  return nhwc_to_ncm_out<float>(src, ld, dst, dst_dtype, sn, sc, off, N, C, HW, st);
        case YOLO_BF16: return nhwc_to_ncm_out<bf16_t>(src, ld, dst, dst_dtype, sn, sc, off, N, C, HW, st);
        case YOLO_F16:  return nhwc_to_ncm_out<f16_t>(src, ld, dst, dst_dtype, sn, sc, off, N, C, HW, st);
    }
    return YOLO_ERR_DTYPE;
}

int yolo_copy_channels(const void* src, int ld_src, void* dst, int ld_dst, long npix, int C, int accumulate,
                       int dtype, hipStream_t st) {
    if (npix <= 0 || C <= 0) return YOLO_OK;
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(src, ld_src, C) && vec_ok<T>(dst, ld_dst, C);
        PICK_V(T, ok, {
            int cv = C / V;
            int g = ew_grid(npix * cv);
            if (accumulate)
                hipLaunchKernelGGL((k_copy_channels<T, V, true>), dim3(g), dim3(TPB), 0, st, (const T*)src, ld_src,
                                   (T*)dst, ld_dst, npix, cv);
            else
                hipLaunchKernelGGL((k_copy_channels<T, V, false>), dim3(g), dim3(TPB), 0, st, (const T*)src, ld_src,
                                   (T*)dst, ld_dst, npix, cv);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

// dst[p][c] = src0 + src1 (+ src2 (+ src3)); every operand an NHWC channel slice with its own row stride; dst may be one
// of the sources.  Replaces autograd's gradient accumulation at fan-out points (model_blocks.py:62,92,156,223-224, neck.py:41-44)
int yolo_add_n(const void* s0, int ld0, const void* s1, int ld1, const void* s2, int ld2, const void* s3, int ld3, int nsrc,
               void* dst, int ld_dst, long npix, int C, int dtype, hipStream_t st) {
    if (nsrc < 2 || nsrc > 4 || npix <= 0) return nsrc >= 2 && nsrc <= 4 ? YOLO_OK : YOLO_ERR_ARG;
    AddSrcs a;
    a.p[0] = s0; a.p[1] = s1; a.p[2] = s2; a.p[3] = s3;
    a.ld[0] = ld0; a.ld[1] = ld1; a.ld[2] = ld2; a.ld[3] = ld3;
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(dst, ld_dst, C);
        for (int k = 0; k < nsrc; ++k) ok = ok && vec_ok<T>(a.p[k], a.ld[k], C);
        PICK_V(T, ok, {
            int cv = C / V;
            int g = ew_grid(npix * cv);
            if (nsrc == 2) hipLaunchKernelGGL((k_add_n<T, V, 2>), dim3(g), dim3(TPB), 0, st, a, (T*)dst, ld_dst, npix, cv);
            else if (nsrc == 3) hipLaunchKernelGGL((k_add_n<T, V, 3>), dim3(g), dim3(TPB), 0, st, a, (T*)dst, ld_dst, npix, cv);
            else hipLaunchKernelGGL((k_add_n<T, V, 4>), dim3(g), dim3(TPB), 0, st, a, (T*)dst, ld_dst, npix, cv);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

// number of partial blocks yolo_bn_stats / yolo_bn_act_bwd_reduce will write for this problem
int yolo_reduce_nblk(long npix, int C) {
    // one partial row per workgroup; same balanced sizing as the other row-strided kernels (<= 1024 workgroups)
    int cv = C / 8 > 0 ? C / 8 : 1;
    int tpr = cv < TPB ? cv : TPB, rpb = TPB / tpr;
    long units = (npix + (long)RS_ROWS * rpb - 1) / ((long)RS_ROWS * rpb);
    if (units < 1) units = 1;
    long cap = npix / 32;              // keep the partial rows (2*C floats each) well below the tensor's own bytes
    if (cap > 1024) cap = 1024;
    if (cap < 1) cap = 1;
    long iters = (units + cap - 1) / cap;
    return (int)((units + iters - 1) / iters);
}

static int launch_reduce(int mode, const void* y, int ldy, const void* dout, int ldd, const float* scale,
                         const float* shift, const float* mean, const float* invstd, long npix, int C, int act,
                         int dtype, float* partial, int nblk, hipStream_t st) {
    if (nblk != yolo_reduce_nblk(npix, C)) return YOLO_ERR_ARG;
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(y, ldy, C) && (mode == 0 || vec_ok<T>(dout, ldd, C));
        PICK_V(T, ok, {
            int cv = C / V;
            int tpr = cv < TPB ? cv : TPB;
            dim3 g(nblk, ceil_div(cv, tpr));
            if (mode == 0)
                hipLaunchKernelGGL((k_channel_reduce<T, V, 0>), g, dim3(TPB), 0, st, (const T*)y, ldy, (const T*)nullptr,
                                   0, scale, shift, mean, invstd, npix, C, act, partial);
            else
                hipLaunchKernelGGL((k_channel_reduce<T, V, 1>), g, dim3(TPB), 0, st, (const T*)y, ldy, (const T*)dout,
                                   ldd, scale, shift, mean, invstd, npix, C, act, partial);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_bn_stats(const void* y, int ldy, long npix, int C, int dtype, float* partial, int nblk, hipStream_t st) {
    return launch_reduce(0, y, ldy, nullptr, 0, nullptr, nullptr, nullptr, nullptr, npix, C, 0, dtype, partial, nblk, st);
}

int yolo_bn_finalize(const float* partial, int nblk, long count, int C, const void* gamma, const void* beta,
                     void* running_mean, void* running_var, float momentum, float eps, float* mean,
                     float* invstd, float* scale, float* shift, int pdtype, int bdtype, hipStream_t st) {
    if (!pdt_ok(pdtype) || !pdt_ok(bdtype)) return YOLO_ERR_DTYPE;
    hipLaunchKernelGGL(k_bn_finalize, dim3(ceil_div(C, FIN_CH)), dim3(FIN_CH * FIN_PARTS), 0, st, partial, nblk, (float)count, C,
                       gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, pdtype, bdtype);
    return YOLO_LAUNCH_CHECK();
}

int yolo_bn_eval_coeffs(const void* gamma, const void* beta, const void* running_mean, const void* running_var,
                        float eps, int C, float* scale, float* shift, int pdtype, int bdtype, hipStream_t st) {
    if (!pdt_ok(pdtype) || !pdt_ok(bdtype)) return YOLO_ERR_DTYPE;
    hipLaunchKernelGGL(k_bn_eval_coeffs, dim3(ceil_div(C, 128)), dim3(128), 0, st, gamma, beta, running_mean, running_var, eps, C,
                       scale, shift, pdtype, bdtype);
    return YOLO_LAUNCH_CHECK();
}

int yolo_sum_finalize(const float* partial, int nblk, int C, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_sum_finalize, dim3(ceil_div(C, FIN_CH)), dim3(FIN_CH * FIN_PARTS), 0, st, partial, nblk, C, out);
    return YOLO_LAUNCH_CHECK();
}

int yolo_bn_act_fwd(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldres,
                    void* out, int ldout, long npix, int C, int act, int dtype, hipStream_t st) {
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(y, ldy, C) && vec_ok<T>(out, ldout, C) && (!res || vec_ok<T>(res, ldres, C));
        PICK_V(T, ok, {
            int cv = C / V;
            hipLaunchKernelGGL((k_bn_act_fwd<T, V>), rs_grid(npix, cv), dim3(TPB), 0, st, (const T*)y, ldy, scale,
                               shift, (const T*)res, ldres, (T*)out, ldout, npix, cv, act);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_bn_act_bwd_reduce(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift,
                           const float* mean, const float* invstd, long npix, int C, int act, int dtype,
                           float* partial, int nblk, hipStream_t st) {
    return launch_reduce(1, y, ldy, dout, ldd, scale, shift, mean, invstd, npix, C, act, dtype, partial, nblk, st);
}

int yolo_bn_bwd_finalize(const float* partial, int nblk, long count, int C, const void* gamma, const float* mean,
                         const float* invstd, void* dgamma, void* dbeta, float* coef, int pdtype, hipStream_t st) {
    if (!pdt_ok(pdtype)) return YOLO_ERR_DTYPE;
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(ceil_div(C, FIN_CH)), dim3(FIN_CH * FIN_PARTS), 0, st, partial, nblk,
                       (float)count, C, gamma, mean, invstd, dgamma, dbeta, coef, pdtype);
    return YOLO_LAUNCH_CHECK();
}

int yolo_bn_act_bwd_apply(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* coef, void* dy, int lddy, long npix,
                          int C, int act, int dtype, hipStream_t st) {
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(y, ldy, C) && vec_ok<T>(dout, ldd, C) && vec_ok<T>(dy, lddy, C);
        PICK_V(T, ok, {
            int cv = C / V;
            const int tpr = cv < TPB ? cv : TPB;
            hipLaunchKernelGGL((k_bn_act_bwd_apply<T, V>), rs_grid(npix, cv), dim3(TPB), 5 * tpr * V * sizeof(float), st, (const T*)dout,
                               ldd, (const T*)y, ldy, scale, shift, mean, invstd, coef, (T*)dy, lddy, npix, C, act);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

// ---- accumulator form (no finalize launches).  acc: fp32 [YOLO_BN_REPL = 8][2][C], zeroed by the caller.
int yolo_bn_acc_elems(int C) { return BN_REPL * 2 * C; }

static int launch_acc(int mode, const void* y, int ldy, const void* dout, int ldd, const float* gamma, const float* beta,
                      const float* mean, const float* invstd, long npix, int C, int act, int dtype, float* acc,
                      hipStream_t st) {
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(y, ldy, C) && (mode == 0 || vec_ok<T>(dout, ldd, C));
        PICK_V(T, ok, {
            const RsPlan pl = rs_plan(npix, C / V);
            if (mode == 0)
                hipLaunchKernelGGL((k_channel_acc<T, V, 0, 0>), pl.grid, dim3(TPB), 0, st, (const T*)y, ldy, (const T*)nullptr, 0,
                                   gamma, beta, mean, invstd, npix, C, act, acc, pl.tpr, (bn_rev() >> (mode ? 1 : 0)) & 1);
            else if (act)     // the activation is a template parameter: no per-element select between SiLU and identity
                hipLaunchKernelGGL((k_channel_acc<T, V, 1, 1>), pl.grid, dim3(TPB), 0, st, (const T*)y, ldy, (const T*)dout, ldd,
                                   gamma, beta, mean, invstd, npix, C, act, acc, pl.tpr, (bn_rev() >> (mode ? 1 : 0)) & 1);
            else
                hipLaunchKernelGGL((k_channel_acc<T, V, 1, 0>), pl.grid, dim3(TPB), 0, st, (const T*)y, ldy, (const T*)dout, ldd,
                                   gamma, beta, mean, invstd, npix, C, act, acc, pl.tpr, (bn_rev() >> (mode ? 1 : 0)) & 1);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_bn_stats_acc(const void* y, int ldy, long npix, int C, int dtype, float* acc, hipStream_t st) {
    return launch_acc(0, y, ldy, nullptr, 0, nullptr, nullptr, nullptr, nullptr, npix, C, 0, dtype, acc, st);
}

int yolo_bn_finalize_acc(const float* acc, long count, int C, const void* gamma, const void* beta, void* running_mean,
                         void* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                         float* shift, int pdtype, int bdtype, hipStream_t st) {
    if (!pdt_ok(pdtype) || !pdt_ok(bdtype)) return YOLO_ERR_DTYPE;
    hipLaunchKernelGGL(k_bn_finalize_acc, dim3(ceil_div(C, 64)), dim3(64), 0, st, acc, (float)count, C, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift, pdtype, bdtype);
    return YOLO_LAUNCH_CHECK();
}

int yolo_bn_act_fwd_train(const void* y, int ldy, const float* acc, long count, const void* gamma, const void* beta,
                          void* running_mean, void* running_var, float momentum, float eps, float* mean, float* invstd,
                          float* scale, float* shift, const void* res, int ldres, void* out, int ldout, long npix, int C,
                          int act, int dtype, int pdtype, int bdtype, hipStream_t st) {
    if (!pdt_ok(pdtype) || !pdt_ok(bdtype)) return YOLO_ERR_DTYPE;
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(y, ldy, C) && vec_ok<T>(out, ldout, C) && (!res || vec_ok<T>(res, ldres, C));
        PICK_V(T, ok, {
            const RsPlan pl = rs_plan(npix, C / V);
            if (act)
                hipLaunchKernelGGL((k_bn_act_fwd_train<T, V, 1>), pl.grid, dim3(TPB), 2 * pl.tpr * V * sizeof(float), st,
                               (const T*)y, ldy, acc, (float)count, gamma, beta, running_mean, running_var, momentum, eps,
                               mean, invstd, scale, shift, (const T*)res, ldres, (T*)out, ldout, npix, C, act, pl.tpr, pdtype, bdtype, bn_rev() & 1);
            else
                hipLaunchKernelGGL((k_bn_act_fwd_train<T, V, 0>), pl.grid, dim3(TPB), 2 * pl.tpr * V * sizeof(float), st,
                               (const T*)y, ldy, acc, (float)count, gamma, beta, running_mean, running_var, momentum, eps,
                               mean, invstd, scale, shift, (const T*)res, ldres, (T*)out, ldout, npix, C, act, pl.tpr, pdtype, bdtype, bn_rev() & 1);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

// backward pass 1 with atomics: acc[8][2][C] (zeroed by the caller) += (sum dz, sum dz*y), dz = dout*act'(y*scale+shift)
int yolo_bn_bwd_reduce_acc(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift,
                           long npix, int C, int act, int dtype, float* acc, hipStream_t st) {
    return launch_acc(1, y, ldy, dout, ldd, scale, shift, nullptr, nullptr, npix, C, act, dtype, acc, st);
}

int yolo_bn_act_bwd_apply_train(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift,
                                const void* gamma, const float* mean, const float* invstd, const float* acc, long count,
                                void* dgamma, void* dbeta, void* dy, int lddy, long npix, int C, int act, int dtype,
                                int pdtype, hipStream_t st) {
    if (!pdt_ok(pdtype)) return YOLO_ERR_DTYPE;
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(y, ldy, C) && vec_ok<T>(dout, ldd, C) && vec_ok<T>(dy, lddy, C);
        PICK_V(T, ok, {
            const RsPlan pl = rs_plan(npix, C / V);
            if (act)
                hipLaunchKernelGGL((k_bn_act_bwd_apply_train<T, V, 1>), pl.grid, dim3(TPB), 5 * pl.tpr * V * sizeof(float), st,
                               (const T*)dout, ldd, (const T*)y, ldy, scale, shift, gamma, mean, invstd, acc, (float)count,
                               dgamma, dbeta, (T*)dy, lddy, npix, C, act, pl.tpr, pdtype, (bn_rev() >> 2) & 1);
            else
                hipLaunchKernelGGL((k_bn_act_bwd_apply_train<T, V, 0>), pl.grid, dim3(TPB), 5 * pl.tpr * V * sizeof(float), st,
                               (const T*)dout, ldd, (const T*)y, ldy, scale, shift, gamma, mean, invstd, acc, (float)count,
                               dgamma, dbeta, (T*)dy, lddy, npix, C, act, pl.tpr, pdtype, (bn_rev() >> 2) & 1);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_maxpool5_fwd(const void* x, int ldx, void* out, int ldo, uint8_t* idx, int N, int H, int W, int C,
                      int dtype, hipStream_t st) {
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(x, ldx, C) && vec_ok<T>(out, ldo, C);
        PICK_V(T, ok, {
            hipLaunchKernelGGL((k_maxpool5_fwd<T, V>), dim3(ew_grid((long)N * H * W * (C / V))), dim3(TPB), 0, st,
                               (const T*)x, ldx, (T*)out, ldo, idx, N, H, W, C);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_maxpool5_bwd(const void* dout, int ldd, const uint8_t* idx, void* dx, int ldx, int N, int H, int W, int C,
                      int accumulate, int dtype, hipStream_t st) {
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(dout, ldd, C) && vec_ok<T>(dx, ldx, C);
        PICK_V(T, ok, {
            int g = ew_grid((long)N * H * W * (C / V));
            if (accumulate)
                hipLaunchKernelGGL((k_maxpool5_bwd<T, V, true>), dim3(g), dim3(TPB), 0, st, (const T*)dout, ldd, idx,
                                   (T*)dx, ldx, N, H, W, C);
            else
                hipLaunchKernelGGL((k_maxpool5_bwd<T, V, false>), dim3(g), dim3(TPB), 0, st, (const T*)dout, ldd, idx,
                                   (T*)dx, ldx, N, H, W, C);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_upsample2x_fwd(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, int dtype,
                        hipStream_t st) {
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(x, ldx, C) && vec_ok<T>(out, ldo, C);
        PICK_V(T, ok, {
            hipLaunchKernelGGL((k_upsample2x_fwd<T, V>), dim3(ew_grid((long)N * 4 * H * W * (C / V))), dim3(TPB), 0, st,
                               (const T*)x, ldx, (T*)out, ldo, N, H, W, C);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

int yolo_upsample2x_bwd(const void* dout, int ldd, void* dx, int ldx, int N, int H, int W, int C, int accumulate,
                        int dtype, hipStream_t st) {
    YOLO_DISPATCH_T(dtype, {
        bool ok = vec_ok<T>(dout, ldd, C) && vec_ok<T>(dx, ldx, C);
        PICK_V(T, ok, {
            int g = ew_grid((long)N * H * W * (C / V));
            if (accumulate)
                hipLaunchKernelGGL((k_upsample2x_bwd<T, V, true>), dim3(g), dim3(TPB), 0, st, (const T*)dout, ldd,
                                   (T*)dx, ldx, N, H, W, C);
            else
                hipLaunchKernelGGL((k_upsample2x_bwd<T, V, false>), dim3(g), dim3(TPB), 0, st, (const T*)dout, ldd,
                                   (T*)dx, ldx, N, H, W, C);
        });
    });
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
