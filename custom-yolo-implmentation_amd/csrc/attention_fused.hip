// PSA attention core for 16-bit tensors with dk = 32, dh = 64: training up to 448 tokens (every preset at 640x640: 400
// tokens), forward-only passes at any length (k_attn_fwd_long below),
// flash-style on the matrix cores: no score / probability / dP matrix ever reaches memory (the batched-GEMM route of
// attention.hip writes and re-reads 85 MB of fp32 scores, 42 MB of probabilities and the same again for dP / dS per
// step, in 14 launches; it remains for longer sequences and other head shapes).
//
// Everything is computed TRANSPOSED, keys on the rows: S^T = K Q^T.  A 16x16 tile of S^T leaves the MFMA with a lane
// holding keys 4g..4g+3 (g = lane / 16) of query column lane % 16 -- and that IS the B-operand layout of the next
// product (8 consecutive k per lane for one column) once two key tiles are paired: k index g*8 + e  <->  key
// 32u + 4g + e (e < 4) | 32u + 16 + 4g + (e - 4).  The A operand of that product (V^T, K^T, dO^T or Q^T: "k" = token
// rows of a [token][channel] LDS image) comes out of the transposing LDS read with the same pairing (rows 32u + 4g + 0..3
// and 32u + 16 + 4g + 0..3).  So P goes from the softmax to O = P V, and dS to dQ / dK, without leaving registers.
// A workgroup is up to 16 waves x 16 tokens of one (image, head) -- 400 tokens: two workgroups of 13 waves, 256 workgroups
// for 32 images x 4 heads = one per CU -- and stages what it needs of the head in LDS once (K and V, or Q and dO).
//   forward   S^T tiles are recomputed rather than kept (one MFMA each): pass 1 column maxima, pass 2 p = exp(s - max)
//             summed and fed unnormalised to O^T = V^T p^T, O divided by the sum at the end; row log-sum-exp kept
//   backward  k_attn_bwd_dq (per query block): D = rowsum(dO o O), P recomputed from the log-sum-exp, dP^T = V dO^T,
//             dS = P (dP - D) scale, dQ^T = K^T dS^T;   k_attn_bwd_dkv (per key block): S = Q K^T, P, dP = dO V^T, dS,
//             dV^T = dO^T P (+ the gradient of the re-gathered v), dK^T = Q^T dS
// Reference: src/model/model_blocks.py:186-197.
#include <cstdlib>
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <typename T> struct mm;
template <> struct mm<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct mm<f16_t> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// element e of the result (lane i16 of its 16-lane group) = LDS[row(p_lo) + e][col(p_lo) + i16 - 4*(i16 & 3) ...]: the
// hardware transposes a 4-row x 16-column block per lane group; lo gives k elements 0..3, hi 4..7 (as in wgrad_mfma.hip)
template <typename T>
__device__ __forceinline__ typename mm<T>::frag tr_frag(const T* p_lo, const T* p_hi) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_lo);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_hi);
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename mm<T>::frag, both);
}

template <typename T>
__device__ __forceinline__ typename mm<T>::frag load_frag(const T* p, bool ok) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (ok) v = *reinterpret_cast<const uint4*>(p);
    return __builtin_bit_cast(typename mm<T>::frag, v);
}

template <typename T>
__device__ __forceinline__ typename mm<T>::frag pack_frag(const float (&v)[8]) {
    typename mm<T>::frag f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = from_f<T>(v[e]);
    return f;
}

__device__ __forceinline__ float xmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xsum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

struct FDims { int N, T, heads, ldq, ldo, ldv; float scale; };

constexpr int DK = 32, DH = 64, CQ = 2 * DK + DH;
constexpr int LDV = 72, LDK = 32;             // LDS row strides (elements) of 64- and 32-channel token images (wgrad_mfma.hip's)

// rows [0, rows) x `width` channels starting at column `col0` of this (image, head)'s slice -> LDS image, zeros past T.
// Eight 16-byte loads per thread in flight (clamped addresses, no branch between them): the first form, one load and one
// LDS store per loop trip, paid a memory round trip per trip -- 21 trips for the 86 KB a workgroup stages.
template <typename T, int LD>
__device__ __forceinline__ void stage_tokens(T* dst, const T* __restrict__ base, long row_stride, int col0, int width, int rows, int Tn) {
    const int cpr = width / 8, total = rows * cpr, nthr = blockDim.x;
    constexpr int U = 8;
    for (int i0 = threadIdx.x; i0 < total; i0 += nthr * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + u * nthr;
            const int r = idx / cpr, ch = (idx - r * cpr) * 8;
            const int rc = (idx < total && r < Tn) ? r : 0;
            v[u] = *reinterpret_cast<const uint4*>(base + (long)rc * row_stride + col0 + (idx < total ? ch : 0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + u * nthr;
            if (idx < total) {
                const int r = idx / cpr, ch = (idx - r * cpr) * 8;
                *reinterpret_cast<uint4*>(dst + r * LD + ch) = r < Tn ? v[u] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
}

// One workgroup = blockDim / 64 waves x 16 tokens of one (image, head).  K and V of the head sit in LDS; S^T tiles are
// recomputed instead of kept (one MFMA each): pass 1 finds the column maxima, pass 2 forms p = exp(s - max), sums it and
// feeds O^T = V^T p^T with the unnormalised p; O is divided by the sum at the end.
template <typename T, int NPAIR>
__global__ __launch_bounds__(1024) void k_attn_fwd_fused(FDims a, const T* __restrict__ qkv, T* __restrict__ o, T* __restrict__ vp,
                                                         float* __restrict__ lse) {
    typedef typename mm<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ROWS = NPAIR * 32;
    T* Ks = reinterpret_cast<T*>(smem_raw);                   // [ROWS][LDK]
    T* Vs = Ks + ROWS * LDK;                                  // [ROWS][LDV]
    const int qb = (blockDim.x >> 6) * 16;
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * qb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, c4 = 4 * (i16 & 3);
    const T* base = qkv + (long)n * a.T * a.ldq + h * CQ;
    stage_tokens<T, LDK>(Ks, base, a.ldq, DK, DK, ROWS, a.T);
    stage_tokens<T, LDV>(Vs, base, a.ldq, 2 * DK, DH, ROWS, a.T);
    const int qrow = q0 + wave * 16 + i16;
    const bool qok = qrow < a.T;
    const frag bq = load_frag<T>(base + (long)(qok ? qrow : 0) * a.ldq + g * 8, true);
    __syncthreads();
    // v re-gathered to [token][head*dh] for the positional depthwise conv: this workgroup's tokens, from the LDS image
    for (int idx = threadIdx.x; idx < qb * (DH / 8); idx += blockDim.x) {
        const int r = idx / (DH / 8), ch = (idx - r * (DH / 8)) * 8;
        if (q0 + r < a.T)
            *reinterpret_cast<uint4*>(vp + ((long)n * a.T + q0 + r) * a.ldv + h * DH + ch) = *reinterpret_cast<const uint4*>(Vs + (q0 + r) * LDV + ch);
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2 * NPAIR; ++t) {
        const f32x4 st = mm<T>::mma(*reinterpret_cast<const frag*>(Ks + (16 * t + i16) * LDK + g * 8), bq, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * t + 4 * g + r < a.T) m = fmaxf(m, st[r] * a.scale);
    }
    m = xmax(m);
    __asm__ volatile("" ::: "memory");                        // pass 2 re-reads K: keeping 26 score tiles alive would cost 104 registers
    float l = 0.f;
    f32x4 oacc[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NPAIR; ++u) {
        float pv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t = 2 * u + half;
            const f32x4 st = mm<T>::mma(*reinterpret_cast<const frag*>(Ks + (16 * t + i16) * LDK + g * 8), bq, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (16 * t + 4 * g + r < a.T) ? __expf(st[r] * a.scale - m) : 0.f;
                pv[half * 4 + r] = p;
                l += p;
            }
        }
        const frag pb = pack_frag<T>(pv);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
            const T* p = Vs + (32 * u + 4 * g + q) * LDV + dt * 16 + c4;
            oacc[dt] = mm<T>::mma(tr_frag<T>(p, p + 16 * LDV), pb, oacc[dt]);
        }
    }
    l = xsum(l);
    const float inv = 1.f / l;
    if (qok) {
        T* orow = o + ((long)n * a.T + qrow) * a.ldo + h * DH;
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = oacc[dt][r] * inv;
            store_pack<T, 4>(orow + dt * 16 + 4 * g, v);
        }
        if (g == 0 && lse != nullptr) lse[((long)n * a.heads + h) * a.T + qrow] = m + __logf(l);
    }
}

// Forward without a backward stash for sequences beyond one LDS image (1280 x 1280 inputs: 1600 tokens; inference and
// no-grad passes).  The keys are walked in blocks of 256 tokens -- K and V of one block in LDS -- with the running column
// maximum and sum rescaled per block (online softmax); inside a block the two passes of k_attn_fwd_fused.  The batched-GEMM
// route it replaces writes 41 MB of fp32 scores and 20 MB of probabilities per image of preset l and takes ~100 us per
// PSABlock at 4 images.
constexpr int LONG_NP = 8, LONG_ROWS = LONG_NP * 32;
template <typename T>
__global__ __launch_bounds__(1024) void k_attn_fwd_long(FDims a, const T* __restrict__ qkv, T* __restrict__ o, T* __restrict__ vp) {
    typedef typename mm<T>::frag frag;
    __shared__ __attribute__((aligned(16))) T Ks[LONG_ROWS * LDK];
    __shared__ __attribute__((aligned(16))) T Vs[LONG_ROWS * LDV];
    const int qb = (blockDim.x >> 6) * 16;
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * qb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, c4 = 4 * (i16 & 3);
    const T* base = qkv + (long)n * a.T * a.ldq + h * CQ;
    const int qrow = q0 + wave * 16 + i16;
    const bool qok = qrow < a.T;
    const frag bq = load_frag<T>(base + (long)(qok ? qrow : 0) * a.ldq + g * 8, true);
    // v re-gathered to [token][head*dh] for the positional depthwise conv: this workgroup's tokens, straight from qkv
    for (int idx = threadIdx.x; idx < qb * (DH / 8); idx += blockDim.x) {
        const int r = idx / (DH / 8), ch = (idx - r * (DH / 8)) * 8;
        if (q0 + r < a.T)
            *reinterpret_cast<uint4*>(vp + ((long)n * a.T + q0 + r) * a.ldv + h * DH + ch) =
                *reinterpret_cast<const uint4*>(base + (long)(q0 + r) * a.ldq + 2 * DK + ch);
    }
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.T; k0 += LONG_ROWS) {
        __syncthreads();                                      // the previous block's fragment reads are done
        stage_tokens<T, LDK>(Ks, base + (long)k0 * a.ldq, a.ldq, DK, DK, LONG_ROWS, a.T - k0);
        stage_tokens<T, LDV>(Vs, base + (long)k0 * a.ldq, a.ldq, 2 * DK, DH, LONG_ROWS, a.T - k0);
        __syncthreads();
        const int left = a.T - k0;                            // valid keys in this block: >= 1
        float mb = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * LONG_NP; ++t) {
            const f32x4 st = mm<T>::mma(*reinterpret_cast<const frag*>(Ks + (16 * t + i16) * LDK + g * 8), bq, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * t + 4 * g + r < left) mb = fmaxf(mb, st[r] * a.scale);
        }
        const float mn = fmaxf(m, xmax(mb));                  // finite from the first block on
        const float corr = __expf(m - mn);                    // first block: exp(-inf) = 0 on zero accumulators
        m = mn;
        l *= corr;
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= corr;
        __asm__ volatile("" ::: "memory");                    // pass 2 re-reads K instead of keeping 16 score tiles alive
#pragma unroll
        for (int u = 0; u < LONG_NP; ++u) {
            float pv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 2 * u + half;
                const f32x4 st = mm<T>::mma(*reinterpret_cast<const frag*>(Ks + (16 * t + i16) * LDK + g * 8), bq, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = (16 * t + 4 * g + r < left) ? __expf(st[r] * a.scale - m) : 0.f;
                    pv[half * 4 + r] = p;
                    l += p;
                }
            }
            const frag pb = pack_frag<T>(pv);
#pragma unroll
            for (int dt = 0; dt < DH / 16; ++dt) {
                const T* p = Vs + (32 * u + 4 * g + q) * LDV + dt * 16 + c4;
                oacc[dt] = mm<T>::mma(tr_frag<T>(p, p + 16 * LDV), pb, oacc[dt]);
            }
        }
    }
    l = xsum(l);
    const float inv = 1.f / l;
    if (qok) {
        T* orow = o + ((long)n * a.T + qrow) * a.ldo + h * DH;
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = oacc[dt][r] * inv;
            store_pack<T, 4>(orow + dt * 16 + 4 * g, v);
        }
    }
}

// dQ of the workgroup's queries; also D = rowsum(dO o O) for k_attn_bwd_dkv.  K and V of the head in LDS.
template <typename T, int NPAIR>
__global__ __launch_bounds__(1024) void k_attn_bwd_dq(FDims a, const T* __restrict__ qkv, const T* __restrict__ o,
                                                      const T* __restrict__ d_o, int lddo, const float* __restrict__ lse,
                                                      float* __restrict__ Dws, T* __restrict__ dqkv, int lddq) {
    typedef typename mm<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ROWS = NPAIR * 32;
    T* Ks = reinterpret_cast<T*>(smem_raw);                   // [ROWS][LDK]
    T* Vs = Ks + ROWS * LDK;                                  // [ROWS][LDV]
    const int qb = (blockDim.x >> 6) * 16;
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * qb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, c4 = 4 * (i16 & 3);
    const T* base = qkv + (long)n * a.T * a.ldq + h * CQ;
    stage_tokens<T, LDK>(Ks, base, a.ldq, DK, DK, ROWS, a.T);
    stage_tokens<T, LDV>(Vs, base, a.ldq, 2 * DK, DH, ROWS, a.T);
    const int qrow = q0 + wave * 16 + i16;
    const bool qok = qrow < a.T;
    const int qc = qok ? qrow : 0;
    const frag bq = load_frag<T>(base + (long)qc * a.ldq + g * 8, true);
    const T* dorow = d_o + ((long)n * a.T + qc) * lddo + h * DH;
    const T* orow = o + ((long)n * a.T + qc) * a.ldo + h * DH;
    frag bdo[2];
    float dpart = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bdo[kk] = load_frag<T>(dorow + kk * 32 + g * 8, true);
        const frag of = load_frag<T>(orow + kk * 32 + g * 8, true);
#pragma unroll
        for (int e = 0; e < 8; ++e) dpart += to_f<T>(bdo[kk][e]) * to_f<T>(of[e]);
    }
    const float Dq = xsum(dpart);
    const float lq = lse[((long)n * a.heads + h) * a.T + qc];
    if (qok && g == 0) Dws[((long)n * a.heads + h) * a.T + qrow] = Dq;
    __syncthreads();
    f32x4 dqacc[DK / 16];
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt) dqacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NPAIR; ++u) {
        float dsv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t = 2 * u + half;
            const f32x4 st = mm<T>::mma(*reinterpret_cast<const frag*>(Ks + (16 * t + i16) * LDK + g * 8), bq, (f32x4){0.f, 0.f, 0.f, 0.f});
            const T* vr = Vs + (16 * t + i16) * LDV + g * 8;
            f32x4 dpt = mm<T>::mma(*reinterpret_cast<const frag*>(vr), bdo[0], (f32x4){0.f, 0.f, 0.f, 0.f});
            dpt = mm<T>::mma(*reinterpret_cast<const frag*>(vr + 32), bdo[1], dpt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = qok && (16 * t + 4 * g + r < a.T);
                const float p = ok ? __expf(st[r] * a.scale - lq) : 0.f;
                dsv[half * 4 + r] = p * (dpt[r] - Dq) * a.scale;
            }
        }
        const frag dsb = pack_frag<T>(dsv);
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
            const T* p = Ks + (32 * u + 4 * g + q) * LDK + dt * 16 + c4;
            dqacc[dt] = mm<T>::mma(tr_frag<T>(p, p + 16 * LDK), dsb, dqacc[dt]);
        }
    }
    if (qok) {
        T* drow = dqkv + ((long)n * a.T + qrow) * lddq + h * CQ;
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = dqacc[dt][r];
            store_pack<T, 4>(drow + dt * 16 + 4 * g, v);
        }
    }
}

// dK, dV of the workgroup's keys.  Q and dO of the head (and the rows' log-sum-exp and D) in LDS.
template <typename T, int NPAIR>
__global__ __launch_bounds__(1024) void k_attn_bwd_dkv(FDims a, const T* __restrict__ qkv, const T* __restrict__ d_o, int lddo,
                                                       const T* __restrict__ d_vp, int lddv, const float* __restrict__ lse,
                                                       const float* __restrict__ Dws, T* __restrict__ dqkv, int lddq) {
    typedef typename mm<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ROWS = NPAIR * 32;
    T* Qs = reinterpret_cast<T*>(smem_raw);                   // [ROWS][LDK]
    T* dOs = Qs + ROWS * LDK;                                 // [ROWS][LDV]
    float* ls = reinterpret_cast<float*>(dOs + ROWS * LDV);   // [ROWS] log-sum-exp, [ROWS] D
    float* Ds = ls + ROWS;
    const int kb = (blockDim.x >> 6) * 16;
    const int n = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * kb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, c4 = 4 * (i16 & 3);
    const T* base = qkv + (long)n * a.T * a.ldq + h * CQ;
    const T* dobase = d_o + (long)n * a.T * lddo + h * DH;
    stage_tokens<T, LDK>(Qs, base, a.ldq, 0, DK, ROWS, a.T);
    stage_tokens<T, LDV>(dOs, dobase, lddo, 0, DH, ROWS, a.T);
    for (int r = threadIdx.x; r < ROWS; r += blockDim.x) {
        const bool ok = r < a.T;
        ls[r] = ok ? lse[((long)n * a.heads + h) * a.T + r] : 0.f;
        Ds[r] = ok ? Dws[((long)n * a.heads + h) * a.T + r] : 0.f;
    }
    const int krow = k0 + wave * 16 + i16;
    const bool kok = krow < a.T;
    const T* kr = base + (long)(kok ? krow : 0) * a.ldq;
    const frag bk = load_frag<T>(kr + DK + g * 8, true);
    frag bv[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) bv[kk] = load_frag<T>(kr + 2 * DK + kk * 32 + g * 8, true);
    __syncthreads();
    f32x4 dvacc[DH / 16], dkacc[DK / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) dvacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt) dkacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NPAIR; ++u) {
        float pv[8], dsv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t = 2 * u + half;
            // A operands straight from the LDS images (rows = queries 16t + i16, 8 consecutive channels per lane group)
            const frag aq = *reinterpret_cast<const frag*>(Qs + (16 * t + i16) * LDK + g * 8);
            const f32x4 st = mm<T>::mma(aq, bk, (f32x4){0.f, 0.f, 0.f, 0.f});
            const T* dor = dOs + (16 * t + i16) * LDV + g * 8;
            f32x4 dp = mm<T>::mma(*reinterpret_cast<const frag*>(dor), bv[0], (f32x4){0.f, 0.f, 0.f, 0.f});
            dp = mm<T>::mma(*reinterpret_cast<const frag*>(dor + 32), bv[1], dp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = 16 * t + 4 * g + r;
                const bool ok = kok && qi < a.T;
                const float p = ok ? __expf(st[r] * a.scale - ls[qi]) : 0.f;
                pv[half * 4 + r] = p;
                dsv[half * 4 + r] = p * (dp[r] - Ds[qi]) * a.scale;
            }
        }
        const frag pb = pack_frag<T>(pv), dsb = pack_frag<T>(dsv);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
            const T* p = dOs + (32 * u + 4 * g + q) * LDV + dt * 16 + c4;
            dvacc[dt] = mm<T>::mma(tr_frag<T>(p, p + 16 * LDV), pb, dvacc[dt]);
        }
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
            const T* p = Qs + (32 * u + 4 * g + q) * LDK + dt * 16 + c4;
            dkacc[dt] = mm<T>::mma(tr_frag<T>(p, p + 16 * LDK), dsb, dkacc[dt]);
        }
    }
    if (kok) {
        T* drow = dqkv + ((long)n * a.T + krow) * lddq + h * CQ;
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = dkacc[dt][r];
            store_pack<T, 4>(drow + DK + dt * 16 + 4 * g, v);
        }
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = dvacc[dt][r];
            if (d_vp != nullptr) {
                float e[4];
                load_pack<T, 4>(d_vp + ((long)n * a.T + krow) * lddv + h * DH + dt * 16 + 4 * g, e);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += e[r];
            }
            store_pack<T, 4>(drow + 2 * DK + dt * 16 + 4 * g, v);
        }
    }
}

// token blocks per (image, head) and waves per workgroup: as few workgroups as 16 waves allow, equal shares
void blocking(int Tn, int& nblk, int& waves) {
    const int tiles = (Tn + 15) / 16;
    nblk = (tiles + 15) / 16;
    waves = (tiles + nblk - 1) / nblk;
}

int npair_for(int Tn) { return Tn <= 128 ? 4 : Tn <= 224 ? 7 : Tn <= 416 ? 13 : 14; }

template <typename F> int opt_in_lds(F* fn, size_t bytes) {
    if (bytes <= 48 * 1024) return YOLO_OK;
    return hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

template <typename T, int NPAIR>
int fwd_go(const FDims& a, const T* qkv, T* o, T* vp, float* lse, hipStream_t st) {
    const size_t lds = (size_t)NPAIR * 32 * (LDK + LDV) * sizeof(T);
    int rc = opt_in_lds(k_attn_fwd_fused<T, NPAIR>, lds);
    if (rc) return rc;
    int nblk, waves;
    blocking(a.T, nblk, waves);
    hipLaunchKernelGGL((k_attn_fwd_fused<T, NPAIR>), dim3(nblk, a.heads, a.N), dim3(64 * waves), lds, st, a, qkv, o, vp, lse);
    return YOLO_LAUNCH_CHECK();
}

template <typename T, int NPAIR>
int bwd_go(const FDims& a, const T* qkv, const T* o, const T* d_o, int lddo, const T* d_vp, int lddv, const float* lse, float* Dws,
           T* dqkv, int lddq, hipStream_t st) {
    const size_t lds_q = (size_t)NPAIR * 32 * (LDK + LDV) * sizeof(T);
    const size_t lds_kv = (size_t)NPAIR * 32 * (LDK + LDV) * sizeof(T) + (size_t)NPAIR * 32 * 2 * sizeof(float);
    int rc = opt_in_lds(k_attn_bwd_dq<T, NPAIR>, lds_q);
    if (!rc) rc = opt_in_lds(k_attn_bwd_dkv<T, NPAIR>, lds_kv);
    if (rc) return rc;
    int nblk, waves;
    blocking(a.T, nblk, waves);
    const dim3 grid(nblk, a.heads, a.N), block(64 * waves);
    hipLaunchKernelGGL((k_attn_bwd_dq<T, NPAIR>), grid, block, lds_q, st, a, qkv, o, d_o, lddo, lse, Dws, dqkv, lddq);
    hipLaunchKernelGGL((k_attn_bwd_dkv<T, NPAIR>), grid, block, lds_kv, st, a, qkv, d_o, lddo, d_vp, lddv, lse, Dws, dqkv, lddq);
    return YOLO_LAUNCH_CHECK();
}

#define NPAIR_DISPATCH(CALL)                     \
    switch (npair_for(a.T)) {                    \
        case 4: return CALL(4);                  \
        case 7: return CALL(7);                  \
        case 13: return CALL(13);                \
        default: return CALL(14);                \
    }

template <typename T>
int fwd_t(const FDims& a, const void* qkv, void* o, void* vp, float* lse, hipStream_t st) {
#define FWD_CALL(NP) fwd_go<T, NP>(a, (const T*)qkv, (T*)o, (T*)vp, lse, st)
    NPAIR_DISPATCH(FWD_CALL)
#undef FWD_CALL
}

template <typename T>
int bwd_t(const FDims& a, const void* qkv, const void* o, const void* d_o, int lddo, const void* d_vp, int lddv, const float* lse,
          float* Dws, void* dqkv, int lddq, hipStream_t st) {
#define BWD_CALL(NP) bwd_go<T, NP>(a, (const T*)qkv, (const T*)o, (const T*)d_o, lddo, (const T*)d_vp, lddv, lse, Dws, (T*)dqkv, lddq, st)
    NPAIR_DISPATCH(BWD_CALL)
#undef BWD_CALL
}

}  // namespace

// 1 if the fused kernels take this problem (16-bit, dk 32, dh 64, <= 448 tokens, 16-byte aligned rows)
int attn_fused_ok(int T_, int dk, int dh, int dtype) {
    static const int on = [] { const char* e = getenv("YOLO_ATTN_FUSED"); return e ? atoi(e) : 1; }();
    return on && (dtype == YOLO_BF16 || dtype == YOLO_F16) && dk == DK && dh == DH && T_ >= 1 && T_ <= 448;
}

// lse: fp32 [N][heads][T] (the stash kept for the backward)
int attn_fused_fwd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, float* lse, int N, int T_, int heads, float scale,
                   int dtype, hipStream_t st) {
    if (ldq % 8 || ldo % 4 || ldv % 8 || (reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(vp) & 15) ||
        (reinterpret_cast<uintptr_t>(o) & 7))
        return YOLO_ERR_ARG;
    const FDims a{N, T_, heads, ldq, ldo, ldv, scale};
    return dtype == YOLO_BF16 ? fwd_t<bf16_t>(a, qkv, o, vp, lse, st) : fwd_t<f16_t>(a, qkv, o, vp, lse, st);
}

// forward only (no stash): any sequence length; the one-image kernel up to 448 tokens, the key-blocked one beyond
int attn_fused_fwd_nograd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, int N, int T_, int heads, float scale,
                          int dtype, hipStream_t st) {
    if (T_ <= 448) return attn_fused_fwd(qkv, ldq, o, ldo, vp, ldv, nullptr, N, T_, heads, scale, dtype, st);
    if (ldq % 8 || ldo % 4 || ldv % 8 || (reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(vp) & 15) ||
        (reinterpret_cast<uintptr_t>(o) & 7))
        return YOLO_ERR_ARG;
    const FDims a{N, T_, heads, ldq, ldo, ldv, scale};
    // 256 queries per workgroup; 128 when that leaves CUs without one (every workgroup stages all of K and V once)
    const int waves = (long)N * heads * ((T_ + 255) / 256) >= 256 ? 16 : 8;
    const dim3 grid((T_ + waves * 16 - 1) / (waves * 16), heads, N), block(64 * waves);
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((k_attn_fwd_long<bf16_t>), grid, block, 0, st, a, (const bf16_t*)qkv, (bf16_t*)o, (bf16_t*)vp);
    else
        hipLaunchKernelGGL((k_attn_fwd_long<f16_t>), grid, block, 0, st, a, (const f16_t*)qkv, (f16_t*)o, (f16_t*)vp);
    return YOLO_LAUNCH_CHECK();
}

// Dws: fp32 [N][heads][T] scratch; dqkv fully written
int attn_fused_bwd(const void* qkv, int ldq, const void* o, int ldo, const void* d_o, int lddo, const void* d_vp, int lddv,
                   const float* lse, float* Dws, void* dqkv, int lddq, int N, int T_, int heads, float scale, int dtype,
                   hipStream_t st) {
    if (ldq % 8 || ldo % 8 || lddo % 8 || lddq % 4 || (d_vp && lddv % 4) || (reinterpret_cast<uintptr_t>(qkv) & 15) ||
        (reinterpret_cast<uintptr_t>(o) & 15) || (reinterpret_cast<uintptr_t>(d_o) & 15) || (reinterpret_cast<uintptr_t>(dqkv) & 7))
        return YOLO_ERR_ARG;
    const FDims a{N, T_, heads, ldq, ldo, 0, scale};
    return dtype == YOLO_BF16 ? bwd_t<bf16_t>(a, qkv, o, d_o, lddo, d_vp, lddv, lse, Dws, dqkv, lddq, st)
                              : bwd_t<f16_t>(a, qkv, o, d_o, lddo, d_vp, lddv, lse, Dws, dqkv, lddq, st);
}
