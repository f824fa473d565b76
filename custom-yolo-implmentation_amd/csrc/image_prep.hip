// On-device input pipeline: the reference's training transform
//   ToImage -> RandomHorizontalFlip -> Resize((S, S)) -> ColorJitter(brightness, contrast, saturation, hue) ->
//   ToDtype(float32, scale=True) -> Normalize(mean, std)                       (src/data/transforms.py:4-14)
// for a whole batch of decoded uint8 HWC images of different sizes, with the random decisions (flip, the four jitter
// factors and their order) drawn on the host and passed in.  torchvision is not in this image (parity unpinned for
// its arithmetic); the kernels follow torchvision.transforms.v2.functional 0.24 as published:
//   resize        bilinear with antialiasing = torch's _upsample_bilinear2d_aa: separable triangle filter whose support
//                 is the scale factor when shrinking, accumulated in fp32 and rounded to uint8 (the reference resizes
//                 the uint8 image; its fixed-point two-pass form may differ from this by one level)
//   brightness    blend(img, 0, f)            = trunc(clamp(img * f, 0, 255))
//   contrast      blend(img, mean(floor(gray)), f),  gray = 0.2989 R + 0.587 G + 0.114 B
//   saturation    blend(img, floor(gray), f)
//   hue           RGB/255 -> HSV, h = (h + f) mod 1, -> RGB, -> uint8 by floor(x * (256 - 1e-3))
//   to float      x / 255, (x - mean) / std
// Stages: resize+flip -> uint8 planar staging; per jitter slot k = 0..3 a per-image gray-mean reduction and the op that
// image has in slot k; the last launch converts, normalises and writes NCHW in the model's input dtype.
#include "common.h"

struct PrepImage {
    long off;           // byte offset of the image (uint8 HWC, row stride W*3) in the source buffer
    int H, W;
    int flip;
    int order[4];       // op in jitter slot k: 0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none
    float factor[4];    // factor of op o (indexed by op, not by slot)
};

namespace {

__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

// weights of output index i along one axis (torch aten/src/ATen/native/cpu/UpSampleKernel.cpp, antialias bilinear)
struct Taps { int lo, n; float scale, inv, center; };
__device__ __forceinline__ Taps taps_of(int i, int in_size, int out_size) {
    Taps t;
    t.scale = (float)in_size / (float)out_size;
    const float support = t.scale >= 1.f ? t.scale : 1.f;
    t.inv = t.scale >= 1.f ? 1.f / t.scale : 1.f;
    t.center = t.scale * ((float)i + 0.5f);
    t.lo = max(0, (int)(t.center - support + 0.5f));
    t.n = min(in_size, (int)(t.center + support + 0.5f)) - t.lo;
    return t;
}

// one thread per output pixel: all three channels
__global__ __launch_bounds__(256) void k_resize_flip(const unsigned char* __restrict__ src, const PrepImage* __restrict__ imgs,
                                                    unsigned char* __restrict__ stage, int S) {
    const int n = blockIdx.z;
    const PrepImage im = imgs[n];
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    if (ox >= S) return;
    const Taps ty = taps_of(oy, im.H, S), tx = taps_of(ox, im.W, S);
    float wsum_y = 0.f, wsum_x = 0.f;
    for (int j = 0; j < ty.n; ++j) wsum_y += tri(((float)(j + ty.lo) - ty.center + 0.5f) * ty.inv);
    for (int j = 0; j < tx.n; ++j) wsum_x += tri(((float)(j + tx.lo) - tx.center + 0.5f) * tx.inv);
    float acc[3] = {0.f, 0.f, 0.f};
    const unsigned char* base = src + im.off;
    for (int jy = 0; jy < ty.n; ++jy) {
        const float wy = tri(((float)(jy + ty.lo) - ty.center + 0.5f) * ty.inv) / wsum_y;
        const unsigned char* row = base + (long)(ty.lo + jy) * im.W * 3;
        float r[3] = {0.f, 0.f, 0.f};
        for (int jx = 0; jx < tx.n; ++jx) {
            const float wx = tri(((float)(jx + tx.lo) - tx.center + 0.5f) * tx.inv);
            const int sx = tx.lo + jx;
            const unsigned char* p = row + (long)(im.flip ? im.W - 1 - sx : sx) * 3;
            r[0] += wx * (float)p[0]; r[1] += wx * (float)p[1]; r[2] += wx * (float)p[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += wy * r[c] / wsum_x;
    }
    // a flipped image is resized from the flipped source: output column ox reads mirrored columns, i.e. the same
    // result as flipping first (the filter is symmetric); written at ox
#pragma unroll
    for (int c = 0; c < 3; ++c)
        stage[(((long)n * 3 + c) * S + oy) * S + ox] = (unsigned char)fminf(fmaxf(floorf(acc[c] + 0.5f), 0.f), 255.f);
}

__device__ __forceinline__ float gray_of(float r, float g, float b) { return floorf(0.2989f * r + 0.587f * g + 0.114f * b); }

// mean of floor(gray) per image -> means[n] (fp32, zeroed by the caller); one block row per image
__global__ __launch_bounds__(256) void k_gray_mean(const unsigned char* __restrict__ stage, int S, float* __restrict__ means) {
    const int n = blockIdx.y;
    const long hw = (long)S * S;
    const unsigned char* r = stage + (long)n * 3 * hw;
    float acc = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x)
        acc += gray_of((float)r[i], (float)r[hw + i], (float)r[2 * hw + i]);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(means + n, acc / (float)hw);
}

__device__ __forceinline__ float blend_u8(float a, float b, float ratio) {      // trunc(clamp(a*ratio + b*(1-ratio)))
    return floorf(fminf(fmaxf(a * ratio + b * (1.f - ratio), 0.f), 255.f));
}

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float f) {
    // torchvision _rgb_to_hsv / _hsv_to_rgb on [0,1] floats
    r *= (1.f / 255.f); g *= (1.f / 255.f); b *= (1.f / 255.f);
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const bool eqc = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eqc ? 1.f : maxc);
    const float crd = eqc ? 1.f : cr;
    const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
    const float hr = maxc == r ? bc - gc : 0.f;
    const float hg = (maxc == g && maxc != r) ? 2.f + rc - bc : 0.f;
    const float hb = (maxc != g && maxc != r) ? 4.f + gc - rc : 0.f;
    float h = fmodf((hr + hg + hb) / 6.f + 1.f, 1.f);
    h = h + f;
    h = h - floorf(h);                                       // remainder(1.0)
    const float v = maxc;
    const float h6 = h * 6.f;
    const float fi = floorf(h6);
    const float ff = h6 - fi;
    const int i = ((int)fi) % 6;
    const float p = fminf(fmaxf(v * (1.f - s), 0.f), 1.f), q = fminf(fmaxf(v * (1.f - s * ff), 0.f), 1.f);
    const float t = fminf(fmaxf(v * (1.f - s * (1.f - ff)), 0.f), 1.f);
    float ro, go, bo;
    switch (i) {
        case 0: ro = v; go = t; bo = p; break;
        case 1: ro = q; go = v; bo = p; break;
        case 2: ro = p; go = v; bo = t; break;
        case 3: ro = p; go = q; bo = v; break;
        case 4: ro = t; go = p; bo = v; break;
        default: ro = v; go = p; bo = q; break;
    }
    const float k = 255.f + 1.f - 1e-3f;                     // float -> uint8 of to_dtype(scale=True)
    r = floorf(ro * k); g = floorf(go * k); b = floorf(bo * k);
}

// jitter slot `slot` of every image, in place on the staging buffer; FINAL: instead write (x/255 - mean)/std as T NCHW
template <typename T, bool FINAL>
__global__ __launch_bounds__(256) void k_color(unsigned char* __restrict__ stage, const PrepImage* __restrict__ imgs, int slot,
                                               const float* __restrict__ means, int S, T* __restrict__ out, float m0, float m1,
                                               float m2, float s0, float s1, float s2) {
    const int n = blockIdx.y;
    const PrepImage im = imgs[n];
    const int op = slot >= 0 ? im.order[slot] : -1;
    const float f = op >= 0 ? im.factor[op] : 1.f;
    const float mean = means != nullptr ? means[n] : 0.f;
    const long hw = (long)S * S;
    unsigned char* base = stage + (long)n * 3 * hw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
        float r = (float)base[i], g = (float)base[hw + i], b = (float)base[2 * hw + i];
        if (op == 0) { r = blend_u8(r, 0.f, f); g = blend_u8(g, 0.f, f); b = blend_u8(b, 0.f, f); }
        else if (op == 1) { r = blend_u8(r, mean, f); g = blend_u8(g, mean, f); b = blend_u8(b, mean, f); }
        else if (op == 2) { const float gr = gray_of(r, g, b); r = blend_u8(r, gr, f); g = blend_u8(g, gr, f); b = blend_u8(b, gr, f); }
        else if (op == 3) hue_shift(r, g, b, f);
        if (FINAL) {
            T* o = out + (long)n * 3 * hw;
            o[i] = from_f<T>((r * (1.f / 255.f) - m0) / s0);
            o[hw + i] = from_f<T>((g * (1.f / 255.f) - m1) / s1);
            o[2 * hw + i] = from_f<T>((b * (1.f / 255.f) - m2) / s2);
        } else {
            base[i] = (unsigned char)r; base[hw + i] = (unsigned char)g; base[2 * hw + i] = (unsigned char)b;
        }
    }
}

}  // namespace

extern "C" {

int yolo_prep_image_bytes(void) { return (int)sizeof(PrepImage); }

// fill record `index` of the host-side table (order[k] = op of jitter slot k or -1; factor indexed by op)
int yolo_prep_image_fill(void* table_host, int index, long off, int H, int W, int flip, int o0, int o1, int o2, int o3,
                         float brightness, float contrast, float saturation, float hue) {
    if (H <= 0 || W <= 0) return YOLO_ERR_ARG;
    PrepImage& p = ((PrepImage*)table_host)[index];
    p.off = off; p.H = H; p.W = W; p.flip = flip;
    p.order[0] = o0; p.order[1] = o1; p.order[2] = o2; p.order[3] = o3;
    p.factor[0] = brightness; p.factor[1] = contrast; p.factor[2] = saturation; p.factor[3] = hue;
    return YOLO_OK;
}

// src: all images of the batch, uint8 HWC, back to back (table[i].off); stage: uint8 scratch [N][3][S][S]; means: fp32
// scratch [4][N]; out: [N][3][S][S] of out_dtype.  jitter = 0 skips the colour stages (validation transform).
int yolo_image_prep(const void* src, const void* table_dev, int N, int S, int jitter, void* stage, float* means, void* out,
                    int out_dtype, float m0, float m1, float m2, float s0, float s1, float s2, hipStream_t st) {
    if (N <= 0 || S <= 0) return YOLO_ERR_ARG;
    const PrepImage* imgs = (const PrepImage*)table_dev;
    hipLaunchKernelGGL(k_resize_flip, dim3(ceil_div(S, 256), S, N), dim3(256), 0, st, (const unsigned char*)src, imgs,
                       (unsigned char*)stage, S);
    const long hw = (long)S * S;
    int gx = (int)((hw + 255) / 256);
    if (gx > 64) gx = 64;
    if (jitter) {
        int rc = yolo_zero_async(means, (size_t)4 * N * sizeof(float), st);
        if (rc) return rc;
    }
    const int nslot = jitter ? 4 : 1;
    for (int k = 0; k < nslot; ++k) {
        const int slot = jitter ? k : -1;
        float* mk = jitter ? means + (long)k * N : nullptr;
        if (jitter) hipLaunchKernelGGL(k_gray_mean, dim3(gx, N), dim3(256), 0, st, (const unsigned char*)stage, S, mk);
        if (k + 1 < nslot) {
            hipLaunchKernelGGL((k_color<float, false>), dim3(gx, N), dim3(256), 0, st, (unsigned char*)stage, imgs, slot, mk, S,
                               (float*)nullptr, m0, m1, m2, s0, s1, s2);
        } else {
            YOLO_DISPATCH_T(out_dtype, hipLaunchKernelGGL((k_color<T, true>), dim3(gx, N), dim3(256), 0, st, (unsigned char*)stage,
                                                          imgs, slot, mk, S, (T*)out, m0, m1, m2, s0, s1, s2));
        }
    }
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
