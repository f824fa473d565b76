// Stem (first layer: 3 -> C, 3x3 stride 2 pad 1, reference src/model/backbone.py:38).
// Cin = 3 cannot feed the MFMA gather (8-channel packets), and a VALU direct conv cost 8 ms per step.
// Instead the image is unfolded ONCE into a K = 27 (+5 zero) column matrix -- reading the NCHW input
// tensor directly, so the NCHW->NHWC conversion disappears too -- and forward / weight-gradient run
// as 1x1 convolutions on the MFMA kernels.  k = ci*9 + kh*3 + kw is exactly the OIHW flattening, so
// weights and weight gradients are plain [Cout][27] views padded to 32.
#include "common.h"

namespace {

template <typename TI, typename TO>
__global__ void k_stem_im2col(const TI* __restrict__ img, TO* __restrict__ col, int N, int H, int W, int OH, int OW) {
    long total = (long)N * OH * OW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        int ow = (int)(p % OW);
        long t = p / OW;
        int oh = (int)(t % OH);
        long n = t / OH;
        float v[32];
#pragma unroll
        for (int k = 27; k < 32; ++k) v[k] = 0.f;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const TI* plane = img + (n * 3 + ci) * (long)H * W;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                int ih = 2 * oh + kh - 1;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    int iw = 2 * ow + kw - 1;
                    bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
                    v[ci * 9 + kh * 3 + kw] = ok ? to_f<TI>(plane[(long)ih * W + iw]) : 0.f;
                }
            }
        }
        TO* o = col + p * 32;
        constexpr int V = vec_of<TO>::N;
#pragma unroll
        for (int k = 0; k < 32; k += V) {
            float w[V];
#pragma unroll
            for (int j = 0; j < V; ++j) w[j] = v[k + j];
            store_pack<TO, V>(o + k, w);
        }
    }
}

template <typename P, typename T>
__global__ void k_stem_pack_w(const P* __restrict__ w, int Cout, T* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cout * 32) return;
    int o = e >> 5, k = e & 31;
    out[e] = from_f<T>(k < 27 ? to_f<P>(w[o * 27 + k]) : 0.f);
}

template <typename P>
__global__ void k_stem_unpack_dw(const float* __restrict__ dw32, int Cout, P* __restrict__ dw) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cout * 27) return;
    int o = e / 27, k = e - o * 27;
    dw[e] = from_f<P>(dw32[o * 32 + k]);
}

}  // namespace

extern "C" {

// img: (N,3,H,W) contiguous NCHW of img_dtype; col: [N*OH*OW][32] of col_dtype (an NHWC tensor with C = 32)
int yolo_stem_im2col(const void* img, int img_dtype, void* col, int col_dtype, int N, int H, int W, int OH, int OW,
                     hipStream_t st) {
    if (OH != (H - 1) / 2 + 1 || OW != (W - 1) / 2 + 1) return YOLO_ERR_ARG;
    long total = (long)N * OH * OW;
    int grid = (int)((total + 255) / 256 > 256 * 16 ? 256 * 16 : (total + 255) / 256);
    if (grid < 1) grid = 1;
#define STEM_LAUNCH(TI) YOLO_DISPATCH_T(col_dtype, hipLaunchKernelGGL((k_stem_im2col<TI, T>), dim3(grid), dim3(256), 0, st, (const TI*)img, (T*)col, N, H, W, OH, OW))
    switch (img_dtype) {
        case YOLO_F32:  STEM_LAUNCH(float); break;
        case YOLO_BF16: STEM_LAUNCH(bf16_t); break;
        case YOLO_F16:  STEM_LAUNCH(f16_t); break;
        default: return YOLO_ERR_DTYPE;
    }
#undef STEM_LAUNCH
    return YOLO_LAUNCH_CHECK();
}

// w: OIHW (Cout,3,3,3) -> out [Cout][32] (the forward-packed matrix of a 1x1 conv with Cin = 32)
int yolo_stem_pack_weights(const void* w, int w_dtype, int Cout, void* out, int out_dtype, hipStream_t st) {
    int grid = (Cout * 32 + 255) / 256;
#define PACK_LAUNCH(P) YOLO_DISPATCH_T(out_dtype, hipLaunchKernelGGL((k_stem_pack_w<P, T>), dim3(grid), dim3(256), 0, st, (const P*)w, Cout, (T*)out))
    switch (w_dtype) {
        case YOLO_F32:  PACK_LAUNCH(float); break;
        case YOLO_BF16: PACK_LAUNCH(bf16_t); break;
        case YOLO_F16:  PACK_LAUNCH(f16_t); break;
        default: return YOLO_ERR_DTYPE;
    }
#undef PACK_LAUNCH
    return YOLO_LAUNCH_CHECK();
}

// dw32: fp32 (Cout,32,1,1) gradient of the padded 1x1 weights -> dw OIHW (Cout,3,3,3) of dw_dtype
int yolo_stem_unpack_wgrad(const float* dw32, int Cout, void* dw, int dw_dtype, hipStream_t st) {
    int grid = (Cout * 27 + 255) / 256;
    switch (dw_dtype) {
        case YOLO_F32:  hipLaunchKernelGGL((k_stem_unpack_dw<float>), dim3(grid), dim3(256), 0, st, dw32, Cout, (float*)dw); break;
        case YOLO_BF16: hipLaunchKernelGGL((k_stem_unpack_dw<bf16_t>), dim3(grid), dim3(256), 0, st, dw32, Cout, (bf16_t*)dw); break;
        case YOLO_F16:  hipLaunchKernelGGL((k_stem_unpack_dw<f16_t>), dim3(grid), dim3(256), 0, st, dw32, Cout, (f16_t*)dw); break;
        default: return YOLO_ERR_DTYPE;
    }
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
