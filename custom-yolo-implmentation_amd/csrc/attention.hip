// PSA attention core (reference src/model/model_blocks.py:186-197) on NHWC tokens.
//   qkv[n][tok][h*(2dk+dh) + {q:0..dk, k:dk..2dk, v:2dk..2dk+dh}]
//   o[n][i][h*dh+d] = sum_j softmax_j(scale * q_i.k_j) * v_j[d];   vp = v gathered to [n][tok][h*dh+d]
// Token counts are small (400 @640, 1600 @1280) so this is a flash-style fp32 VALU kernel: key/value
// (or query/dO) chunks of 128 rows staged in LDS as fp32 with +1 padded rows, one wave per row,
// online softmax; lse saved for the backward, which recomputes P (two kernels, no atomics:
// A = dQ and D=rowsum(dO*O) per query block, B = dK/dV per key block).
#include "common.h"
#include "gemm.h"

namespace {

constexpr int CH = 128;     // rows per staged chunk
constexpr int QB = 32;      // rows owned by a workgroup (8 per wave)
constexpr int MAXDK = 64, MAXDH = 128;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

struct AttnDims { int N, T, heads, dk, dh, ldq, ldo, ldv; float scale; };

template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ base, long row_stride, int col0, int ncols,
                                           int r0, int nrows_valid, float* __restrict__ dst, int dst_ld) {
    // rows r0.. of `base` (columns col0..col0+ncols) -> dst[r][c]; rows >= nrows_valid zero-filled
    for (int e = threadIdx.x; e < CH * ncols; e += blockDim.x) {
        int r = e / ncols, c = e - r * ncols;
        dst[r * dst_ld + c] = (r < nrows_valid) ? to_f<T>(base[(long)(r0 + r) * row_stride + col0 + c]) : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_attn_fwd(AttnDims a, const T* __restrict__ qkv, T* __restrict__ o,
                                                  T* __restrict__ vp, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dk = a.dk, dh = a.dh, ldk = dk + 1, ldh = dh + 1, cq = 2 * dk + dh;
    float* Ks = sm;                       // [CH][dk+1]
    float* Vs = Ks + CH * ldk;            // [CH][dh+1]
    float* Qs = Vs + CH * ldh;            // [QB][dk+1]
    float* Ps = Qs + QB * ldk;            // [4][CH]
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* base = qkv + (long)n * a.T * a.ldq + h * cq;

    for (int e = threadIdx.x; e < QB * dk; e += blockDim.x) {
        int r = e / dk, c = e - r * dk;
        Qs[r * ldk + c] = (q0 + r < a.T) ? to_f<T>(base[(long)(q0 + r) * a.ldq + c]) : 0.f;
    }
    // v gather for this query block's rows
    for (int e = threadIdx.x; e < QB * dh; e += blockDim.x) {
        int r = e / dh, c = e - r * dh;
        if (q0 + r < a.T) vp[((long)n * a.T + q0 + r) * a.ldv + h * dh + c] = base[(long)(q0 + r) * a.ldq + 2 * dk + c];
    }

    float m[8], l[8], acc[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) { m[r] = -INFINITY; l[r] = 0.f; acc[r][0] = acc[r][1] = 0.f; }

    for (int c0 = 0; c0 < a.T; c0 += CH) {
        int nv = a.T - c0 < CH ? a.T - c0 : CH;
        __syncthreads();
        stage_rows<T>(base, a.ldq, dk, dk, c0, nv, Ks, ldk);
        stage_rows<T>(base, a.ldq, 2 * dk, dh, c0, nv, Vs, ldh);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int qi = wave * 8 + r;
            if (q0 + qi >= a.T) continue;               // wave-uniform
            float s0 = 0.f, s1 = 0.f;
            for (int d = 0; d < dk; ++d) {
                float qd = Qs[qi * ldk + d];
                s0 = fmaf(qd, Ks[lane * ldk + d], s0);
                s1 = fmaf(qd, Ks[(lane + 64) * ldk + d], s1);
            }
            s0 = lane < nv ? s0 * a.scale : -INFINITY;
            s1 = lane + 64 < nv ? s1 * a.scale : -INFINITY;
            float mn = fmaxf(m[r], wave_max(fmaxf(s0, s1)));
            float alpha = __expf(m[r] - mn);
            float p0 = __expf(s0 - mn), p1 = __expf(s1 - mn);
            l[r] = l[r] * alpha + wave_sum(p0 + p1);
            m[r] = mn;
            __builtin_amdgcn_wave_barrier();
            Ps[wave * CH + lane] = p0;
            Ps[wave * CH + lane + 64] = p1;
            __builtin_amdgcn_wave_barrier();
            float a0 = acc[r][0] * alpha, a1 = acc[r][1] * alpha;
            for (int j = 0; j < nv; ++j) {
                float p = Ps[wave * CH + j];
                if (lane < dh) a0 = fmaf(p, Vs[j * ldh + lane], a0);
                if (lane + 64 < dh) a1 = fmaf(p, Vs[j * ldh + lane + 64], a1);
            }
            acc[r][0] = a0; acc[r][1] = a1;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = q0 + wave * 8 + r;
        if (i >= a.T) continue;
        float inv = 1.f / l[r];
        T* orow = o + ((long)n * a.T + i) * a.ldo + h * dh;
        if (lane < dh) orow[lane] = from_f<T>(acc[r][0] * inv);
        if (lane + 64 < dh) orow[lane + 64] = from_f<T>(acc[r][1] * inv);
        if (lane == 0) lse[((long)n * a.heads + h) * a.T + i] = m[r] + __logf(l[r]);
    }
}

// A: per query block.  dq_i = scale * sum_j dS_ij k_j,  dS = P*(dP - D),  D_i = dO_i.O_i
template <typename T>
__global__ __launch_bounds__(256) void k_attn_bwd_q(AttnDims a, const T* __restrict__ qkv, const T* __restrict__ o,
                                                    const T* __restrict__ d_o, int lddo, const float* __restrict__ lse,
                                                    float* __restrict__ Dbuf, T* __restrict__ dqkv, int lddq) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dk = a.dk, dh = a.dh, ldk = dk + 1, ldh = dh + 1, cq = 2 * dk + dh;
    float* Ks = sm;
    float* Vs = Ks + CH * ldk;
    float* Qs = Vs + CH * ldh;            // [QB][dk+1]
    float* dOs = Qs + QB * ldk;           // [QB][dh+1]
    float* Ps = dOs + QB * ldh;           // [4][CH]
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* base = qkv + (long)n * a.T * a.ldq + h * cq;

    for (int e = threadIdx.x; e < QB * dk; e += blockDim.x) {
        int r = e / dk, c = e - r * dk;
        Qs[r * ldk + c] = (q0 + r < a.T) ? to_f<T>(base[(long)(q0 + r) * a.ldq + c]) : 0.f;
    }
    for (int e = threadIdx.x; e < QB * dh; e += blockDim.x) {
        int r = e / dh, c = e - r * dh;
        dOs[r * ldh + c] = (q0 + r < a.T) ? to_f<T>(d_o[((long)n * a.T + q0 + r) * lddo + h * dh + c]) : 0.f;
    }
    __syncthreads();
    float Dv[8], L[8], dq[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int qi = wave * 8 + r, i = q0 + qi;
        dq[r] = 0.f; Dv[r] = 0.f; L[r] = 0.f;
        if (i >= a.T) continue;
        const T* orow = o + ((long)n * a.T + i) * a.ldo + h * dh;
        float part = 0.f;
        if (lane < dh) part += dOs[qi * ldh + lane] * to_f<T>(orow[lane]);
        if (lane + 64 < dh) part += dOs[qi * ldh + lane + 64] * to_f<T>(orow[lane + 64]);
        Dv[r] = wave_sum(part);
        L[r] = lse[((long)n * a.heads + h) * a.T + i];
        if (lane == 0) Dbuf[((long)n * a.heads + h) * a.T + i] = Dv[r];
    }
    for (int c0 = 0; c0 < a.T; c0 += CH) {
        int nv = a.T - c0 < CH ? a.T - c0 : CH;
        __syncthreads();
        stage_rows<T>(base, a.ldq, dk, dk, c0, nv, Ks, ldk);
        stage_rows<T>(base, a.ldq, 2 * dk, dh, c0, nv, Vs, ldh);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int qi = wave * 8 + r;
            if (q0 + qi >= a.T) continue;
            float s0 = 0.f, s1 = 0.f, g0 = 0.f, g1 = 0.f;
            for (int d = 0; d < dk; ++d) {
                float qd = Qs[qi * ldk + d];
                s0 = fmaf(qd, Ks[lane * ldk + d], s0);
                s1 = fmaf(qd, Ks[(lane + 64) * ldk + d], s1);
            }
            for (int d = 0; d < dh; ++d) {
                float gd = dOs[qi * ldh + d];
                g0 = fmaf(gd, Vs[lane * ldh + d], g0);
                g1 = fmaf(gd, Vs[(lane + 64) * ldh + d], g1);
            }
            float p0 = lane < nv ? __expf(s0 * a.scale - L[r]) : 0.f;
            float p1 = lane + 64 < nv ? __expf(s1 * a.scale - L[r]) : 0.f;
            __builtin_amdgcn_wave_barrier();
            Ps[wave * CH + lane] = p0 * (g0 - Dv[r]);
            Ps[wave * CH + lane + 64] = p1 * (g1 - Dv[r]);
            __builtin_amdgcn_wave_barrier();
            if (lane < dk) {
                float t = dq[r];
                for (int j = 0; j < nv; ++j) t = fmaf(Ps[wave * CH + j], Ks[j * ldk + lane], t);
                dq[r] = t;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = q0 + wave * 8 + r;
        if (i < a.T && lane < dk) dqkv[((long)n * a.T + i) * lddq + h * cq + lane] = from_f<T>(dq[r] * a.scale);
    }
}

// B: per key block.  dk_j = scale * sum_i dS_ij q_i ;  dv_j = sum_i P_ij dO_i (+ d_vp_j)
template <typename T>
__global__ __launch_bounds__(256) void k_attn_bwd_kv(AttnDims a, const T* __restrict__ qkv, const T* __restrict__ d_o,
                                                     int lddo, const T* __restrict__ d_vp, int lddv,
                                                     const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                     T* __restrict__ dqkv, int lddq) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dk = a.dk, dh = a.dh, ldk = dk + 1, ldh = dh + 1, cq = 2 * dk + dh;
    float* Qs = sm;                       // [CH][dk+1]   query chunk
    float* dOs = Qs + CH * ldk;           // [CH][dh+1]
    float* Kb = dOs + CH * ldh;           // [QB][dk+1]   this block's keys
    float* Vb = Kb + QB * ldk;            // [QB][dh+1]
    float* Ls = Vb + QB * ldh;            // [CH]
    float* Ds = Ls + CH;                  // [CH]
    float* Ps = Ds + CH;                  // [4][2][CH]   P and dS per wave
    const int n = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * QB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* base = qkv + (long)n * a.T * a.ldq + h * cq;

    for (int e = threadIdx.x; e < QB * dk; e += blockDim.x) {
        int r = e / dk, c = e - r * dk;
        Kb[r * ldk + c] = (k0 + r < a.T) ? to_f<T>(base[(long)(k0 + r) * a.ldq + dk + c]) : 0.f;
    }
    for (int e = threadIdx.x; e < QB * dh; e += blockDim.x) {
        int r = e / dh, c = e - r * dh;
        Vb[r * ldh + c] = (k0 + r < a.T) ? to_f<T>(base[(long)(k0 + r) * a.ldq + 2 * dk + c]) : 0.f;
    }
    float dkk[8], dv0[8], dv1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) dkk[r] = dv0[r] = dv1[r] = 0.f;

    for (int c0 = 0; c0 < a.T; c0 += CH) {
        int nv = a.T - c0 < CH ? a.T - c0 : CH;
        __syncthreads();
        stage_rows<T>(base, a.ldq, 0, dk, c0, nv, Qs, ldk);
        stage_rows<T>(d_o + (long)n * a.T * lddo + h * dh, lddo, 0, dh, c0, nv, dOs, ldh);
        for (int e = threadIdx.x; e < CH; e += blockDim.x) {
            Ls[e] = e < nv ? lse[((long)n * a.heads + h) * a.T + c0 + e] : 0.f;
            Ds[e] = e < nv ? Dbuf[((long)n * a.heads + h) * a.T + c0 + e] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int kj = wave * 8 + r;
            if (k0 + kj >= a.T) continue;
            float s0 = 0.f, s1 = 0.f, g0 = 0.f, g1 = 0.f;
            for (int d = 0; d < dk; ++d) {
                float kd = Kb[kj * ldk + d];
                s0 = fmaf(kd, Qs[lane * ldk + d], s0);
                s1 = fmaf(kd, Qs[(lane + 64) * ldk + d], s1);
            }
            for (int d = 0; d < dh; ++d) {
                float vd = Vb[kj * ldh + d];
                g0 = fmaf(vd, dOs[lane * ldh + d], g0);
                g1 = fmaf(vd, dOs[(lane + 64) * ldh + d], g1);
            }
            float p0 = lane < nv ? __expf(s0 * a.scale - Ls[lane]) : 0.f;
            float p1 = lane + 64 < nv ? __expf(s1 * a.scale - Ls[lane + 64]) : 0.f;
            float* Pw = Ps + wave * 2 * CH;
            __builtin_amdgcn_wave_barrier();
            Pw[lane] = p0; Pw[lane + 64] = p1;
            Pw[CH + lane] = p0 * (g0 - Ds[lane]);
            Pw[CH + lane + 64] = p1 * (g1 - Ds[lane + 64]);
            __builtin_amdgcn_wave_barrier();
            float t0 = dv0[r], t1 = dv1[r], tk = dkk[r];
            for (int i = 0; i < nv; ++i) {
                float p = Pw[i], ds = Pw[CH + i];
                if (lane < dh) t0 = fmaf(p, dOs[i * ldh + lane], t0);
                if (lane + 64 < dh) t1 = fmaf(p, dOs[i * ldh + lane + 64], t1);
                if (lane < dk) tk = fmaf(ds, Qs[i * ldk + lane], tk);
            }
            dv0[r] = t0; dv1[r] = t1; dkk[r] = tk;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int j = k0 + wave * 8 + r;
        if (j >= a.T) continue;
        T* row = dqkv + ((long)n * a.T + j) * lddq + h * cq;
        const T* gv = d_vp ? d_vp + ((long)n * a.T + j) * lddv + h * dh : nullptr;
        if (lane < dk) row[dk + lane] = from_f<T>(dkk[r] * a.scale);
        if (lane < dh) row[2 * dk + lane] = from_f<T>(dv0[r] + (gv ? to_f<T>(gv[lane]) : 0.f));
        if (lane + 64 < dh) row[2 * dk + lane + 64] = from_f<T>(dv1[r] + (gv ? to_f<T>(gv[lane + 64]) : 0.f));
    }
}

bool dims_ok(int dk, int dh) { return dk >= 1 && dk <= MAXDK && dh >= 1 && dh <= MAXDH; }

// dynamic LDS above the 64 KiB default needs the per-function opt-in (gfx950 has 160 KiB per CU)
template <typename F> int allow_lds(F* fn, size_t bytes) {
    if (bytes <= 48 * 1024) return YOLO_OK;
    if (bytes > 160 * 1024) return YOLO_ERR_ARG;
    return hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}


// ------------------------------------------------------------------------------------------------
// bf16 / f16 path: the matmuls run on the matrix cores through the batched GEMM (gemm.h); P is
// materialised ([N*heads][T][Tp], Tp = T rounded up to 32, pad columns zero) and kept for the backward.
// ------------------------------------------------------------------------------------------------
// P = softmax(S) row by row; one wave per row.  Rows of up to 64*RPL columns stay in registers (one read of S, one
// write of P; the first form read every score three times); longer rows take the looped form (RPL = 0).
template <typename T, int RPL>
__global__ void k_softmax_rows(const float* __restrict__ S, T* __restrict__ P, long rows, int Tn, int Tp) {
    const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* s = S + r * Tp;
    T* p = P + r * Tp;
    if constexpr (RPL > 0) {
        float v[RPL];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int j = lane + 64 * i;
            v[i] = j < Tn ? s[j] : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            v[i] = __expf(v[i] - mx);                       // exp(-inf) = 0 for the columns past Tn
            sum += v[i];
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int j = lane + 64 * i;
            if (j < Tp) p[j] = from_f<T>(v[i] * inv);
        }
    } else {
        float mx = -INFINITY;
        for (int j = lane; j < Tn; j += 64) mx = fmaxf(mx, s[j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Tn; j += 64) sum += __expf(s[j] - mx);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int j = lane; j < Tp; j += 64) p[j] = from_f<T>(j < Tn ? __expf(s[j] - mx) * inv : 0.f);
    }
}

// dS = P * (dP - rowsum(P*dP)); one wave per row, same two forms
template <typename T, int RPL>
__global__ void k_softmax_bwd_rows(const T* __restrict__ P, const float* __restrict__ dP, T* __restrict__ dS, long rows,
                                   int Tn, int Tp) {
    const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const T* p = P + r * Tp;
    const float* g = dP + r * Tp;
    T* o = dS + r * Tp;
    if constexpr (RPL > 0) {
        float pv[RPL], gv[RPL];
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int j = lane + 64 * i;
            pv[i] = j < Tn ? to_f<T>(p[j]) : 0.f;
            gv[i] = j < Tn ? g[j] : 0.f;
            d += pv[i] * gv[i];
        }
        d = wave_sum(d);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int j = lane + 64 * i;
            if (j < Tp) o[j] = from_f<T>(pv[i] * (gv[i] - d));
        }
    } else {
        float d = 0.f;
        for (int j = lane; j < Tn; j += 64) d += to_f<T>(p[j]) * g[j];
        d = wave_sum(d);
        for (int j = lane; j < Tp; j += 64) o[j] = from_f<T>(j < Tn ? to_f<T>(p[j]) * (g[j] - d) : 0.f);
    }
}

// dst[p][h*dgs + doff + d] = src[p][h*sgs + soff + d], d < width (16-byte packets)
template <typename T>
__global__ void k_group_copy(const T* __restrict__ src, int lds_, int sgs, int soff, T* __restrict__ dst, int ldd, int dgs,
                             int doff, long npix, int heads, int width) {
    constexpr int V = vec_of<T>::N;
    const int per = heads * (width / V);
    long total = npix * per;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / per;
        int e = (int)(i - p * per);
        int h = e / (width / V), d = (e - h * (width / V)) * V;
        *reinterpret_cast<uint4*>(dst + p * ldd + h * dgs + doff + d) =
            *reinterpret_cast<const uint4*>(src + p * lds_ + h * sgs + soff + d);
    }
}

inline int round32(int t) { return (t + 31) / 32 * 32; }

template <typename T>
int group_copy(const void* src, int lds_, int sgs, int soff, void* dst, int ldd, int dgs, int doff, long npix, int heads,
               int width, hipStream_t st) {
    long total = npix * heads * (width / vec_of<T>::N);
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL((k_group_copy<T>), dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, (const T*)src, lds_, sgs, soff, (T*)dst,
                       ldd, dgs, doff, npix, heads, width);
    return YOLO_LAUNCH_CHECK();
}

GemmOperand operand(const void* p, long b0, long b1, long rs, int kcontig) {
    GemmOperand o;
    o.p = p; o.b0 = b0; o.b1 = b1; o.rs = rs; o.kcontig = kcontig;
    return o;
}

template <typename T>
int attn_fwd_mfma(const T* qkv, int ldq, T* o, int ldo, T* vp, int ldv, T* P, float* S, int N, int Tn, int heads, int dk,
                  int dh, float scale, int dtype, hipStream_t st) {
    const int cq = 2 * dk + dh, Tp = round32(Tn);
    const long pb1 = (long)Tn * Tp, pb0 = pb1 * heads;
    GemmArgs g;
    g.nb0 = N; g.nb1 = heads; g.accumulate = 0;
    // S = scale * Q K^T
    g.A = operand(qkv, (long)Tn * ldq, cq, ldq, 1);
    g.B = operand(qkv + dk, (long)Tn * ldq, cq, ldq, 1);
    g.C = S; g.c_b0 = pb0; g.c_b1 = pb1; g.c_rs = Tp; g.c_f32 = 1; g.alpha = scale;
    g.M = Tn; g.N = Tn; g.K = dk; g.Kvalid = dk;
    int rc = gemm_batched_launch(g, dtype, st);
    if (rc) return rc;
    const long rows = (long)N * heads * Tn;
    if (Tp <= 448)
        hipLaunchKernelGGL((k_softmax_rows<T, 7>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, P, rows, Tn, Tp);
    else
        hipLaunchKernelGGL((k_softmax_rows<T, 0>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, P, rows, Tn, Tp);
    // O = P V
    g.A = operand(P, pb0, pb1, Tp, 1);
    g.B = operand(qkv + 2 * dk, (long)Tn * ldq, cq, ldq, 0);
    g.C = o; g.c_b0 = (long)Tn * ldo; g.c_b1 = dh; g.c_rs = ldo; g.c_f32 = 0; g.alpha = 1.f;
    g.M = Tn; g.N = dh; g.K = Tp; g.Kvalid = Tn;
    rc = gemm_batched_launch(g, dtype, st);
    if (rc) return rc;
    return group_copy<T>(qkv, ldq, cq, 2 * dk, vp, ldv, dh, 0, (long)N * Tn, heads, dh, st);
}

template <typename T>
int attn_bwd_mfma(const T* qkv, int ldq, const T* d_o, int lddo, const T* d_vp, int lddv, const T* P, float* dP, T* dS,
                  T* dqkv, int lddq, int N, int Tn, int heads, int dk, int dh, float scale, int dtype, hipStream_t st) {
    const int cq = 2 * dk + dh, Tp = round32(Tn);
    const long pb1 = (long)Tn * Tp, pb0 = pb1 * heads;
    const long qb0 = (long)Tn * ldq, gb0 = (long)Tn * lddq, ob0 = (long)Tn * lddo;
    GemmArgs g;
    g.nb0 = N; g.nb1 = heads;
    // dP = dO V^T
    g.A = operand(d_o, ob0, dh, lddo, 1);
    g.B = operand(qkv + 2 * dk, qb0, cq, ldq, 1);
    g.C = dP; g.c_b0 = pb0; g.c_b1 = pb1; g.c_rs = Tp; g.c_f32 = 1; g.alpha = 1.f; g.accumulate = 0;
    g.M = Tn; g.N = Tn; g.K = dh; g.Kvalid = dh;
    int rc = gemm_batched_launch(g, dtype, st);
    if (rc) return rc;
    const long rows = (long)N * heads * Tn;
    if (Tp <= 448)
        hipLaunchKernelGGL((k_softmax_bwd_rows<T, 7>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, dS, rows, Tn, Tp);
    else
        hipLaunchKernelGGL((k_softmax_bwd_rows<T, 0>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, dS, rows, Tn, Tp);
    // dV = P^T dO (+ d_vp, copied into the v slots first)
    if (d_vp) {
        rc = group_copy<T>(d_vp, lddv, dh, 0, dqkv, lddq, cq, 2 * dk, (long)N * Tn, heads, dh, st);
        if (rc) return rc;
    }
    g.A = operand(P, pb0, pb1, Tp, 0);
    g.B = operand(d_o, ob0, dh, lddo, 0);
    g.C = dqkv + 2 * dk; g.c_b0 = gb0; g.c_b1 = cq; g.c_rs = lddq; g.c_f32 = 0; g.alpha = 1.f; g.accumulate = d_vp ? 1 : 0;
    g.M = Tn; g.N = dh; g.K = Tp; g.Kvalid = Tn;
    rc = gemm_batched_launch(g, dtype, st);
    if (rc) return rc;
    // dQ = scale * dS K
    g.A = operand(dS, pb0, pb1, Tp, 1);
    g.B = operand(qkv + dk, qb0, cq, ldq, 0);
    g.C = dqkv; g.alpha = scale; g.accumulate = 0;
    g.M = Tn; g.N = dk; g.K = Tp; g.Kvalid = Tn;
    rc = gemm_batched_launch(g, dtype, st);
    if (rc) return rc;
    // dK = scale * dS^T Q
    g.A = operand(dS, pb0, pb1, Tp, 0);
    g.B = operand(qkv, qb0, cq, ldq, 0);
    g.C = dqkv + dk;
    return gemm_batched_launch(g, dtype, st);
}

}  // namespace

// attention_fused.hip: flash-style MFMA kernels for dk = 32, dh = 64, <= 448 tokens (stash = row log-sum-exp, fp32)
int attn_fused_ok(int T_, int dk, int dh, int dtype);
int attn_fused_fwd_nograd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, int N, int T_, int heads, float scale,
                          int dtype, hipStream_t st);
int attn_fused_fwd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, float* lse, int N, int T_, int heads, float scale,
                   int dtype, hipStream_t st);
int attn_fused_bwd(const void* qkv, int ldq, const void* o, int ldo, const void* d_o, int lddo, const void* d_vp, int lddv,
                   const float* lse, float* Dws, void* dqkv, int lddq, int N, int T_, int heads, float scale, int dtype,
                   hipStream_t st);

extern "C" {

// the same two sizes for a given head shape: the fused kernels keep only the row log-sum-exp and need one fp32 per row of scratch
size_t yolo_attn_stash_bytes_for(int N, int T_, int heads, int dk, int dh, int dtype) {
    if (dtype != YOLO_F32 && !attn_fused_ok(T_, dk, dh, dtype)) return (size_t)N * heads * T_ * round32(T_) * 2;
    return (size_t)N * heads * T_ * 4;
}
size_t yolo_attn_workspace_bytes_for(int N, int T_, int heads, int dk, int dh, int dtype) {
    if (dtype != YOLO_F32 && !attn_fused_ok(T_, dk, dh, dtype)) return (size_t)N * heads * T_ * round32(T_) * 6;
    return (size_t)N * heads * T_ * 4;
}

// bytes kept from forward to backward (fp32: row log-sum-exp; bf16/f16: the probability matrices)
size_t yolo_attn_stash_bytes(int N, int T_, int heads, int dtype) {
    if (dtype == YOLO_F32) return (size_t)N * heads * T_ * 4;
    return (size_t)N * heads * T_ * round32(T_) * 2;
}

// scratch bytes (either direction)
size_t yolo_attn_workspace_bytes(int N, int T_, int heads, int dtype) {
    if (dtype == YOLO_F32) return (size_t)N * heads * T_ * 4;
    return (size_t)N * heads * T_ * round32(T_) * 6;          // fp32 scores / dP + 16-bit dS
}

// o, vp: (N, T, heads*dh)
int yolo_attn_fwd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, void* stash, void* ws, int N, int T_,
                  int heads, int dk, int dh, float scale, int dtype, hipStream_t st) {
    if (!dims_ok(dk, dh)) return YOLO_ERR_ARG;
    if (dtype != YOLO_F32) {
        if (dk % 8 || dh % 8 || ldq % 8 || ldo % 4 || ldv % 8) return YOLO_ERR_ARG;
        if (attn_fused_ok(T_, dk, dh, dtype))
            return attn_fused_fwd(qkv, ldq, o, ldo, vp, ldv, (float*)stash, N, T_, heads, scale, dtype, st);
        if (dtype == YOLO_BF16)
            return attn_fwd_mfma<bf16_t>((const bf16_t*)qkv, ldq, (bf16_t*)o, ldo, (bf16_t*)vp, ldv, (bf16_t*)stash, (float*)ws, N,
                                         T_, heads, dk, dh, scale, dtype, st);
        return attn_fwd_mfma<f16_t>((const f16_t*)qkv, ldq, (f16_t*)o, ldo, (f16_t*)vp, ldv, (f16_t*)stash, (float*)ws, N, T_,
                                    heads, dk, dh, scale, dtype, st);
    }
    AttnDims a{N, T_, heads, dk, dh, ldq, ldo, ldv, scale};
    size_t smem = sizeof(float) * (size_t)(CH * (dk + 1) + CH * (dh + 1) + QB * (dk + 1) + 4 * CH);
    dim3 grid(ceil_div(T_, QB), heads, N);
    int rc = allow_lds(k_attn_fwd<float>, smem);
    if (rc) return rc;
    hipLaunchKernelGGL((k_attn_fwd<float>), grid, dim3(256), smem, st, a, (const float*)qkv, (float*)o, (float*)vp, (float*)stash);
    return YOLO_LAUNCH_CHECK();
}

// The same forward when no backward will follow (inference, no-grad): nothing is stashed and the fused kernels take any
// sequence length (key-blocked online softmax beyond 448 tokens -- 1600 at 1280 x 1280).  Returns 1 when they do not take
// the problem (fp32, other head shapes, YOLO_ATTN_FUSED=0): nothing was launched, the caller uses yolo_attn_fwd.
int yolo_attn_fwd_nograd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, int N, int T_, int heads, int dk, int dh,
                         float scale, int dtype, hipStream_t st) {
    if (!dims_ok(dk, dh)) return YOLO_ERR_ARG;
    if (dtype == YOLO_F32 || !attn_fused_ok(1, dk, dh, dtype) || T_ < 1) return 1;
    return attn_fused_fwd_nograd(qkv, ldq, o, ldo, vp, ldv, N, T_, heads, scale, dtype, st);
}

// dqkv (N, T, heads*(2dk+dh)) fully written; d_vp may be null
int yolo_attn_bwd(const void* qkv, int ldq, const void* o, int ldo, const void* d_o, int lddo, const void* d_vp, int lddv,
                  const void* stash, void* ws, void* dqkv, int lddq, int N, int T_, int heads, int dk, int dh, float scale,
                  int dtype, hipStream_t st) {
    if (!dims_ok(dk, dh)) return YOLO_ERR_ARG;
    if (dtype != YOLO_F32) {
        if (dk % 8 || dh % 8 || ldq % 8 || lddo % 8 || lddq % 4 || (d_vp && lddv % 8)) return YOLO_ERR_ARG;
        if (attn_fused_ok(T_, dk, dh, dtype))
            return attn_fused_bwd(qkv, ldq, o, ldo, d_o, lddo, d_vp, lddv, (const float*)stash, (float*)ws, dqkv, lddq, N, T_, heads,
                                  scale, dtype, st);
        const size_t n = (size_t)N * heads * T_ * round32(T_);
        float* dP = (float*)ws;
        void* dS = (char*)ws + n * 4;
        if (dtype == YOLO_BF16)
            return attn_bwd_mfma<bf16_t>((const bf16_t*)qkv, ldq, (const bf16_t*)d_o, lddo, (const bf16_t*)d_vp, lddv,
                                         (const bf16_t*)stash, dP, (bf16_t*)dS, (bf16_t*)dqkv, lddq, N, T_, heads, dk, dh, scale,
                                         dtype, st);
        return attn_bwd_mfma<f16_t>((const f16_t*)qkv, ldq, (const f16_t*)d_o, lddo, (const f16_t*)d_vp, lddv,
                                    (const f16_t*)stash, dP, (f16_t*)dS, (f16_t*)dqkv, lddq, N, T_, heads, dk, dh, scale, dtype, st);
    }
    AttnDims a{N, T_, heads, dk, dh, ldq, ldo, 0, scale};
    size_t smA = sizeof(float) * (size_t)(CH * (dk + 1) + CH * (dh + 1) + QB * (dk + 1) + QB * (dh + 1) + 4 * CH);
    size_t smB = sizeof(float) * (size_t)(CH * (dk + 1) + CH * (dh + 1) + QB * (dk + 1) + QB * (dh + 1) + 2 * CH + 8 * CH);
    dim3 grid(ceil_div(T_, QB), heads, N);
    int rc = allow_lds(k_attn_bwd_q<float>, smA);
    if (!rc) rc = allow_lds(k_attn_bwd_kv<float>, smB);
    if (rc) return rc;
    hipLaunchKernelGGL((k_attn_bwd_q<float>), grid, dim3(256), smA, st, a, (const float*)qkv, (const float*)o, (const float*)d_o,
                       lddo, (const float*)stash, (float*)ws, (float*)dqkv, lddq);
    hipLaunchKernelGGL((k_attn_bwd_kv<float>), grid, dim3(256), smB, st, a, (const float*)qkv, (const float*)d_o, lddo,
                       (const float*)d_vp, lddv, (const float*)stash, (const float*)ws, (float*)dqkv, lddq);
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
