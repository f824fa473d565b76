// Batched bf16/f16 MFMA GEMM with free operand layouts (see gemm.h).  64x64 output tile per 256-thread
// workgroup (2x2 waves, 2x2 MFMA 16x16x32 tiles each), K in 32-steps, single LDS stage: the token-level
// matmuls it serves (400..1600 tokens, 32..64 features) are launch-latency sized, not throughput sized.
#include "gemm.h"

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

constexpr int LDK = 40;   // row stride (elements) of a K-contiguous tile [64][32]
constexpr int LDR = 72;   // row stride of a row-contiguous tile [32][64]
constexpr int TILE_ELEMS = 64 * LDK;   // 2560 >= 32 * LDR = 2304

// stage one 64 x 32 operand tile: `rows0` = first row index (m or n), `k0` = first k
template <typename T>
__device__ __forceinline__ void stage(const T* __restrict__ base, const GemmOperand& op, int rows0, int nrows, int k0,
                                      int K, int Kvalid, T* __restrict__ lds) {
    const int tid = threadIdx.x;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (op.kcontig) {
        const int r = tid >> 2, kc = (tid & 3) * 8;
        if (rows0 + r < nrows && k0 + kc < K) v = *reinterpret_cast<const uint4*>(base + (long)(rows0 + r) * op.rs + k0 + kc);
        *reinterpret_cast<uint4*>(lds + r * LDK + kc) = v;
    } else {
        const int kr = tid >> 3, rc = (tid & 7) * 8;
        if (k0 + kr < Kvalid && rows0 + rc < nrows) v = *reinterpret_cast<const uint4*>(base + (long)(k0 + kr) * op.rs + rows0 + rc);
        *reinterpret_cast<uint4*>(lds + kr * LDR + rc) = v;
    }
}

// MFMA operand fragment for 16 rows starting at r0: element j of lane l = X(row r0 + (l&15), k = 8*(l>>4) + j)
template <typename T>
__device__ __forceinline__ typename mm<T>::frag fragment(const T* __restrict__ lds, int kcontig, int r0, int lane) {
    typedef typename mm<T>::frag frag;
    const int i = lane & 15, g = lane >> 4;
    if (kcontig) return *reinterpret_cast<const frag*>(lds + (r0 + i) * LDK + g * 8);
    const T* a1 = lds + (8 * g + (i >> 2)) * LDR + r0 + 4 * (i & 3);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1 + 4 * LDR));
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(frag, both);
}

template <typename T>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) T lds_a[TILE_ELEMS];
    __shared__ __attribute__((aligned(16))) T lds_b[TILE_ELEMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int b0 = blockIdx.z / g.nb1, b1 = blockIdx.z - b0 * g.nb1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const T* A = (const T*)g.A.p + b0 * g.A.b0 + b1 * g.A.b1;
    const T* B = (const T*)g.B.p + b0 * g.B.b0 + b1 * g.B.b1;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < g.K; k0 += 32) {
        __syncthreads();
        stage<T>(A, g.A, m0, g.M, k0, g.K, g.Kvalid, lds_a);
        stage<T>(B, g.B, n0, g.N, k0, g.K, g.Kvalid, lds_b);
        __syncthreads();
        typename mm<T>::frag fm[2], fn[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fm[i] = fragment<T>(lds_a, g.A.kcontig, wm * 32 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fn[j] = fragment<T>(lds_b, g.B.kcontig, wn * 32 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mm<T>::mma(fn[j], fm[i], acc[i][j]);   // rows = n, cols = m
    }
    // lane holds C(m = .. + (lane&15), n = .. + (lane>>4)*4 + r), r = 0..3
    const long cb = b0 * g.c_b0 + b1 * g.c_b1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = g.alpha * acc[i][j][r];
            const long off = cb + (long)m * g.c_rs + n;
            if (n + 4 > g.N) {                      // ragged right edge: element-wise
                for (int r = 0; r < 4 && n + r < g.N; ++r) {
                    if (g.c_f32) {
                        float* c = (float*)g.C + off + r;
                        *c = g.accumulate ? *c + v[r] : v[r];
                    } else {
                        T* c = (T*)g.C + off + r;
                        *c = from_f<T>(g.accumulate ? to_f<T>(*c) + v[r] : v[r]);
                    }
                }
                continue;
            }
            if (g.c_f32) {
                float* c = (float*)g.C + off;
                if (g.accumulate) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += c[r];
                }
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                T* c = (T*)g.C + off;
                if (g.accumulate) {
                    float o[4];
                    load_pack<T, 4>(c, o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += o[r];
                }
                store_pack<T, 4>(c, v);
            }
        }
    }
}

}  // namespace

int gemm_batched_launch(const GemmArgs& g, int dtype, hipStream_t st) {
    if ((g.A.kcontig && g.K % 8) || (g.B.kcontig && g.K % 8) || g.c_rs % 4) return YOLO_ERR_ARG;
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64), g.nb0 * g.nb1);
    if (dtype == YOLO_BF16) hipLaunchKernelGGL((k_gemm<bf16_t>), grid, dim3(256), 0, st, g);
    else if (dtype == YOLO_F16) hipLaunchKernelGGL((k_gemm<f16_t>), grid, dim3(256), 0, st, g);
    else return YOLO_ERR_DTYPE;
    return YOLO_LAUNCH_CHECK();
}
