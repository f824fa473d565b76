// Dtype-generic (f32 / bf16 / f16, fp32 accumulate) convolution kernels: the correctness anchor and
// the path for shapes the MFMA kernels do not take (fp32, stem Cin=3, odd channel counts), plus
// depthwise 3x3, weight packing and the host-side launch logic shared with conv_mfma.hip.
#include "common.h"
#include "conv_geom.h"

extern "C" int yolo_bn_stats_acc(const void* y, int ldy, long npix, int C, int dtype, float* acc, hipStream_t st);

// implemented in conv_mfma.hip
int mfma_conv_eligible(const ConvGeom& g, int dtype, const void* src, const void* wm, const void* dst);
int mfma_conv_plan(const ConvGeom& g, int dtype);
int ring_conv_eligible(const ConvGeom& g, int dtype, const void* src, const void* wm, const void* dst);
int ring_conv_plan(const ConvGeom* gs, int n);
int up2_conv_variant(const ConvGeom* gs, int dtype);
int up2_conv_launch(const ConvGeom* gs, int variant, const long* wm_off, long wm_elems, const void* src, const void* wm, void* dst,
                    int accumulate, int dtype, hipStream_t st);
int ring_conv_launch(const ConvGeom* gs, int n, const long* wm_off, long wm_elems, const void* src, const void* wm,
                     const float* bias, void* dst, int accumulate, int dtype, hipStream_t st);
int mfma_conv_launch(const ConvGeom& g, const void* src, const void* wm, const float* bias, void* dst,
                     int accumulate, int dtype, hipStream_t st);
int mfma_wgrad_eligible(int Cin, int Cout, int ldx, int ldy, int dtype, const void* x, const void* dy);
long mfma_wgrad2_plan(int Kpad, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k);
long mfma_wgrad2_ws_elems(int Kpad, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k);
int mfma_wgrad2_launch(const void* x, int ldx, const void* dy, int ldy, float* part, void* dw_oihw, int dw_dtype, int Kpad,
                       int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, hipStream_t st);
int f32_conv_eligible(const ConvGeom& g, const void* src, const void* wm, const void* dst);
int f32_conv_launch(const ConvGeom& g, const float* src, const float* wm, const float* bias, float* dst, int accumulate,
                    hipStream_t st);
int f32_wgrad_eligible(const void* x, int ldx, const void* dy, int ldy, int Cin, int Cout);
int f32_wgrad_launch(const float* x, int ldx, const float* dy, int ldy, float* dwp, int Kpad, int N, int H, int W, int Cin,
                     int OH, int OW, int Cout, int k, int stride, hipStream_t st);
int mfma_wgrad_launch(const void* x, int ldx, const void* dy, int ldy, float* dwp, int Kpad, int N, int H, int W,
                      int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, hipStream_t st);

namespace {

// ------------------------------------------------------------------------------------------------
// generic gather conv: one thread per (dst pixel, cd), cd fastest
// ------------------------------------------------------------------------------------------------
template <typename T, int V, bool ACC>
__global__ void k_conv_generic(ConvGeom g, const T* __restrict__ src, const T* __restrict__ wm,
                               const float* __restrict__ bias, T* __restrict__ dst) {
    long total = (long)g.N * g.Hg * g.Wg * g.Cd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cd = (int)(i % g.Cd);
        long q = i / g.Cd;
        int b = (int)(q % g.Wg);
        long t2 = q / g.Wg;
        int a = (int)(t2 % g.Hg);
        long n = t2 / g.Hg;
        float acc = bias ? bias[cd] : 0.f;
        const T* wrow = wm + (long)cd * g.Kpad;
        for (int t = 0; t < g.ntaps; ++t) {
            int hs = a * g.sstride + g.dh[t], ws = b * g.sstride + g.dw[t];
            if (hs < 0 || hs >= g.Hs || ws < 0 || ws >= g.Ws) continue;
            const T* sp = src + ((n * g.Hs + hs) * g.Ws + ws) * (long)g.lds;
            const T* wp = wrow + t * g.Cs;
            for (int c = 0; c < g.Cs; c += V) {
                float xs[V], wv[V];
                load_pack<T, V>(sp + c, xs);
                load_pack<T, V>(wp + c, wv);
#pragma unroll
                for (int j = 0; j < V; ++j) acc = fmaf(xs[j], wv[j], acc);
            }
        }
        long dp = ((n * g.Hd + a * g.ostep + g.ooff_h) * g.Wd + b * g.ostep + g.ooff_w) * (long)g.ldd + cd;
        if (ACC) acc += to_f<T>(dst[dp]);
        dst[dp] = from_f<T>(acc);
    }
}

// dst pixels of an odd-sized map that no parity class covers do not exist; but classes with zero
// taps never occur (every class has >= 1 tap), so every dx element is written exactly once.

// ------------------------------------------------------------------------------------------------
// generic wgrad: dwp[co][t*Cin+ci] += sum_p dy[p][co] * x[p (+) t][ci]; slabs of output rows, atomics
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_wgrad_generic(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int ldy,
                                float* __restrict__ dwp, int Kpad, int N, int H, int W, int Cin, int OH, int OW,
                                int Cout, int k, int stride, int rows_per_slab) {
    const int K = k * k * Cin;
    long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= (long)Cout * K) return;
    int co = (int)(e / K);
    int kk = (int)(e - (long)co * K);
    int t = kk / Cin, ci = kk - t * Cin;
    int kh = t / k, kw = t - kh * k, pad = k / 2;
    long r0 = (long)blockIdx.y * rows_per_slab, r1 = r0 + rows_per_slab;
    long nrows = (long)N * OH;
    if (r1 > nrows) r1 = nrows;
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) {
        long n = r / OH;
        int oh = (int)(r - n * OH);
        int ih = oh * stride + kh - pad;
        if (ih < 0 || ih >= H) continue;
        const T* dyr = dy + (r * OW) * (long)ldy + co;
        const T* xr = x + ((n * H + ih) * W) * (long)ldx + ci;
        for (int ow = 0; ow < OW; ++ow) {
            int iw = ow * stride + kw - pad;
            if (iw < 0 || iw >= W) continue;
            acc = fmaf(to_f<T>(dyr[(long)ow * ldy]), to_f<T>(xr[(long)iw * ldx]), acc);
        }
    }
    atomicAdd(dwp + (long)co * Kpad + kk, acc);
}

// ------------------------------------------------------------------------------------------------
// weight packing.  Source: OIHW parameter (dtype P).  Destination rows of Kpad elements of T.
//   fwd  : out[o][t*I + i] = w[o][i][kh_t][kw_t]
//   dgrad: out[i][t*O + o] = w[o][i][kh_t][kw_t]     (kh_t, kw_t from conv_taps)
// ------------------------------------------------------------------------------------------------
struct TapIdx { int n; int kh[9], kw[9]; };

template <typename P, typename T>
__global__ void k_pack_weights(const P* __restrict__ w, int O, int I, int k, int mode, TapIdx taps,
                               T* __restrict__ out, int Kpad) {
    const int rows = mode == 0 ? O : I;
    const int inner = mode == 0 ? I : O;
    long total = (long)rows * Kpad;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int r = (int)(e / Kpad);
        int kk = (int)(e - (long)r * Kpad);
        float v = 0.f;
        if (kk < taps.n * inner) {
            int t = kk / inner, c = kk - t * inner;
            int o = mode == 0 ? r : c, i = mode == 0 ? c : r;
            v = to_f<P>(w[(((long)o * I + i) * k + taps.kh[t]) * k + taps.kw[t]]);
        }
        out[e] = from_f<T>(v);
    }
}

// dw (OIHW, dtype P) = dwp[o][(kh*k+kw)*I + i]
template <typename P>
__global__ void k_unpack_wgrad(const float* __restrict__ dwp, int O, int I, int k, int Kpad, P* __restrict__ dw) {
    long total = (long)O * I * k * k;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int kw = (int)(e % k);
        long r = e / k;
        int kh = (int)(r % k);
        r /= k;
        int i = (int)(r % I);
        int o = (int)(r / I);
        dw[e] = from_f<P>(dwp[(long)o * Kpad + (kh * k + kw) * I + i]);
    }
}

// ------------------------------------------------------------------------------------------------
// depthwise 3x3 stride 1 pad 1.  Weights: fp32 [C][9] (the OIHW parameter (C,1,3,3) viewed flat,
// cast by the caller).  FLIP selects the data-gradient form.
// ------------------------------------------------------------------------------------------------
template <typename T, int V, bool FLIP>
__global__ void k_dw3x3(const T* __restrict__ x, int ldx, const float* __restrict__ w, T* __restrict__ y, int ldy,
                        int N, int H, int W, int C, int accumulate) {
    // row-strided: a thread owns one channel group (its 9 x V taps live in registers), a workgroup walks
    // image rows, threads of one group stride along the row -- no per-element index division
    const int cv = C / V;
    const int tpr = cv < 256 ? cv : 256, rpb = 256 / tpr;
    const int wl = threadIdx.x / tpr;
    const int cg = blockIdx.y * tpr + (threadIdx.x - wl * tpr);
    if (wl >= rpb || cg >= cv) return;
    float wt[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < V; ++j) wt[t][j] = w[(cg * V + j) * 9 + (FLIP ? 8 - t : t)];
    const int nrows = N * H;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / H, hh = row - n * H;
        for (int ww = wl; ww < W; ww += rpb) {
            float acc[V];
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hs = hh + kh - 1;
                if (hs < 0 || hs >= H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ws = ww + kw - 1;
                    if (ws < 0 || ws >= W) continue;
                    float a[V];
                    load_pack<T, V>(x + (((long)n * H + hs) * W + ws) * ldx + cg * V, a);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] = fmaf(a[j], wt[kh * 3 + kw][j], acc[j]);
                }
            }
            if (accumulate) {                             // gradient fan-in: add to what another consumer's backward left
                float o[V];
                load_pack<T, V>(y + ((long)row * W + ww) * ldy + cg * V, o);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += o[j];
            }
            store_pack<T, V>(y + ((long)row * W + ww) * ldy + cg * V, acc);
        }
    }
}

// dw[c][tap] += sum_p dy[p][c] * x[p (+) tap][c]; block = 64 channels x 4 row-lanes, slab of rows
// Same thread mapping as k_dw3x3; each thread keeps 9 x V partial sums, the workgroup combines them
// through LDS one tap at a time and issues one atomic per (channel, tap).
template <typename T, int V>
__global__ void k_dw3x3_wgrad(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int ldy,
                              float* __restrict__ dw, int N, int H, int W, int C, int rows_per_slab) {
    __shared__ float red[256][V + 1];
    const int cv = C / V;
    const int tpr = cv < 256 ? cv : 256, rpb = 256 / tpr;
    const int wl = threadIdx.x / tpr;
    const int cg = blockIdx.y * tpr + (threadIdx.x - wl * tpr);
    const bool active = wl < rpb && cg < cv;
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[t][j] = 0.f;
    const int nrows = N * H;
    int r0 = blockIdx.x * rows_per_slab, r1 = r0 + rows_per_slab;
    if (r1 > nrows) r1 = nrows;
    if (active) {
        for (int row = r0; row < r1; ++row) {
            const int n = row / H, hh = row - n * H;
            for (int ww = wl; ww < W; ww += rpb) {
                float g[V];
                load_pack<T, V>(dy + ((long)row * W + ww) * ldy + cg * V, g);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hs = hh + kh - 1;
                    if (hs < 0 || hs >= H) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int ws = ww + kw - 1;
                        if (ws < 0 || ws >= W) continue;
                        float a[V];
                        load_pack<T, V>(x + (((long)n * H + hs) * W + ws) * ldx + cg * V, a);
#pragma unroll
                        for (int j = 0; j < V; ++j) acc[kh * 3 + kw][j] = fmaf(g[j], a[j], acc[kh * 3 + kw][j]);
                    }
                }
            }
        }
    }
    // per-slab partial sums (no atomics: 9*C addresses would each take one add per slab, the contended
    // regime of the global-atomic unit); k_dw_wgrad_finalize sums the slabs
    float* part = dw + (long)blockIdx.x * C * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < V; ++j) red[threadIdx.x][j] = acc[t][j];
        __syncthreads();
        // one thread per (channel group, lane value): tpr*V sums over the workgroup's rpb row lanes, in parallel
        // (a single lane row walking all of them serially cost tens of microseconds per workgroup on narrow layers)
        for (int o = threadIdx.x; o < tpr * V; o += 256) {
            const int cl = o / V, j = o - cl * V;
            const int cgo = blockIdx.y * tpr + cl;
            float s = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < rpb; ++rr) s += red[rr * tpr + cl][j];
            if (cgo < cv) part[(long)(cgo * V + j) * 9 + t] = s;
        }
    }
}

// out[e] = sum_b part[b][e]; (32 elements x 32 parts) per 1024-thread workgroup
__global__ void k_dw_wgrad_finalize(const float* __restrict__ part, int nslab, int E, float* __restrict__ out) {
    __shared__ float red[32][33];
    const int el = threadIdx.x & 31, pj = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float s = 0.f;
    if (e < E)
        for (int b = pj; b < nslab; b += 32) s += part[(long)b * E + e];
    red[pj][el] = s;
    __syncthreads();
    if (pj == 0 && e < E) {
        float t = 0.f;
        for (int j = 0; j < 32; ++j) t += red[j][el];
        out[e] = t;
    }
}

template <typename T> __host__ bool vecok(const void* p, long ld, int C) {
    constexpr int V = vec_of<T>::N;
    return (C % V == 0) && (ld % V == 0) && ((reinterpret_cast<uintptr_t>(p) % 16) == 0);
}

int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

extern "C" int yolo_copy_channels(const void* src, int ld_src, void* dst, int ld_dst, long npix, int C, int accumulate, int dtype,
                                  hipStream_t st);      // elementwise.hip

int launch_generic(const ConvGeom& g, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                   int dtype, hipStream_t st) {
    long total = (long)g.N * g.Hg * g.Wg * g.Cd;
    if (total == 0) return YOLO_OK;
    if (dtype == YOLO_F32 && f32_conv_eligible(g, src, wm, dst))        // LDS-tiled fp32 kernel (conv_f32.hip)
        return f32_conv_launch(g, (const float*)src, (const float*)wm, bias, (float*)dst, accumulate, st);
    YOLO_DISPATCH_T(dtype, {
        bool ok = vecok<T>(src, g.lds, g.Cs) && vecok<T>(wm, g.Kpad, g.Cs);
        if (ok) {
            constexpr int V = vec_of<T>::N;
            if (accumulate) hipLaunchKernelGGL((k_conv_generic<T, V, true>), dim3(grid_for(total)), dim3(256), 0, st, g,
                                               (const T*)src, (const T*)wm, bias, (T*)dst);
            else hipLaunchKernelGGL((k_conv_generic<T, V, false>), dim3(grid_for(total)), dim3(256), 0, st, g,
                                    (const T*)src, (const T*)wm, bias, (T*)dst);
        } else {
            if (accumulate) hipLaunchKernelGGL((k_conv_generic<T, 1, true>), dim3(grid_for(total)), dim3(256), 0, st, g,
                                               (const T*)src, (const T*)wm, bias, (T*)dst);
            else hipLaunchKernelGGL((k_conv_generic<T, 1, false>), dim3(grid_for(total)), dim3(256), 0, st, g,
                                    (const T*)src, (const T*)wm, bias, (T*)dst);
        }
    });
    return YOLO_LAUNCH_CHECK();
}

int run_conv(const ConvGeom& g, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
             int dtype, int algo, hipStream_t st) {
    if (algo != 1 && mfma_conv_eligible(g, dtype, src, wm, dst))
        return mfma_conv_launch(g, src, wm, bias, dst, accumulate, dtype, st);
    if (algo == 2) return YOLO_ERR_ARG;   // MFMA demanded but the shape is not eligible
    int rc = launch_generic(g, src, wm, bias, dst, accumulate, dtype, st);
    if (rc == YOLO_OK && accumulate && g.acc2 != nullptr)      // the generic kernels have one accumulate source: add the second
        rc = yolo_copy_channels(g.acc2, g.ld2, dst, g.ldd, (long)g.N * g.Hd * g.Wd, g.Cd, 1, dtype, st);
    if (rc == YOLO_OK && g.stats)         // the generic kernel has no statistics epilogue: one extra pass over y
        rc = yolo_bn_stats_acc(dst, g.ldd, (long)g.N * g.Hd * g.Wd, g.Cd, dtype, g.stats, st);
    return rc;
}

ConvGeom fwd_geom(int ldx, int ldy, float* stats_acc, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride) {
    ConvGeom g;
    g.stats = stats_acc;
    g.acc2 = nullptr; g.ld2 = 0;
    g.act = 0; g.res = nullptr; g.ldr = 0;
    g.N = N; g.Hs = H; g.Ws = W; g.Cs = Cin; g.lds = ldx;
    g.Hd = OH; g.Wd = OW; g.Cd = Cout; g.ldd = ldy; g.Hg = OH; g.Wg = OW;
    g.ostep = 1; g.ooff_h = 0; g.ooff_w = 0; g.sstride = stride;
    int kh[9], kw[9];
    g.ntaps = conv_taps(0, k, stride, 0, g.dh, g.dw, kh, kw);
    g.K = g.ntaps * Cin; g.Kpad = round_up32(g.K);
    return g;
}

// data gradient, parity class c (stride 2: four classes; stride 1: c = 0)
ConvGeom dgrad_geom(int lddy, int lddx, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int c) {
    ConvGeom g;
    g.stats = nullptr;
    g.acc2 = nullptr; g.ld2 = 0;
    g.act = 0; g.res = nullptr; g.ldr = 0;
    g.N = N; g.Hs = OH; g.Ws = OW; g.Cs = Cout; g.lds = lddy;
    g.Hd = H; g.Wd = W; g.Cd = Cin; g.ldd = lddx;
    int kh[9], kw[9];
    g.ntaps = conv_taps(1, k, stride, c, g.dh, g.dw, kh, kw);
    g.K = g.ntaps * Cout; g.Kpad = round_up32(g.K);
    if (stride == 1) {
        g.Hg = H; g.Wg = W; g.ostep = 1; g.ooff_h = 0; g.ooff_w = 0; g.sstride = 1;
    } else {
        int ph = c >> 1, pw = c & 1;
        g.Hg = ph == 0 ? (H + 1) / 2 : H / 2;
        g.Wg = pw == 0 ? (W + 1) / 2 : W / 2;
        g.ostep = 2; g.ooff_h = ph; g.ooff_w = pw; g.sstride = 1;
    }
    return g;
}

bool supported(int k, int stride) { return (k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)); }

}  // namespace

extern "C" {

// Elements per packed row: K = ntaps * inner rounded up to 32 (zero padded), so MFMA K-steps never
// read past a row.
int yolo_conv_kpad(int O, int I, int k, int stride, int mode, int cls) {
    int dh[9], dw[9], kh[9], kw[9];
    int nt = conv_taps(mode, k, stride, cls, dh, dw, kh, kw);
    return round_up32(nt * (mode == 0 ? I : O));
}

// total elements of the dgrad weight buffer (1 matrix for stride 1, 4 class matrices for stride 2)
long yolo_conv_dgrad_wbuf_elems(int O, int I, int k, int stride) {
    if (stride == 1) return (long)I * yolo_conv_kpad(O, I, k, 1, 1, 0);
    long s = 0;
    for (int c = 0; c < 4; ++c) s += (long)I * yolo_conv_kpad(O, I, k, 2, 1, c);
    return s;
}

// mode 0: forward matrix [O][Kpad].  mode 1: dgrad buffer (all classes, back to back).
int yolo_conv_pack_weights(const void* w_oihw, int w_dtype, int O, int I, int k, int stride, int mode, void* out,
                           int out_dtype, hipStream_t st) {
    if (!supported(k, stride)) return YOLO_ERR_ARG;
    int ncls = (mode == 1 && stride == 2) ? 4 : 1;
    long off = 0;
    for (int c = 0; c < ncls; ++c) {
        TapIdx taps;
        int dh[9], dw[9];
        taps.n = conv_taps(mode, k, stride, c, dh, dw, taps.kh, taps.kw);
        int Kpad = round_up32(taps.n * (mode == 0 ? I : O));
        long total = (long)(mode == 0 ? O : I) * Kpad;
#define PACK_LAUNCH(P)                                                                                               \
    YOLO_DISPATCH_T(out_dtype, hipLaunchKernelGGL((k_pack_weights<P, T>), dim3(grid_for(total)), dim3(256), 0, st,   \
                                                  (const P*)w_oihw, O, I, k, mode, taps, (T*)out + off, Kpad))
        switch (w_dtype) {
            case YOLO_F32:  PACK_LAUNCH(float); break;
            case YOLO_BF16: PACK_LAUNCH(bf16_t); break;
            case YOLO_F16:  PACK_LAUNCH(f16_t); break;
            default: return YOLO_ERR_DTYPE;
        }
#undef PACK_LAUNCH
        off += total;
    }
    return YOLO_LAUNCH_CHECK();
}

int yolo_conv_unpack_wgrad(const float* dwp, int O, int I, int k, void* dw_oihw, int dw_dtype, hipStream_t st) {
    int Kpad = round_up32(k * k * I);
    long total = (long)O * I * k * k;
    switch (dw_dtype) {
        case YOLO_F32:  hipLaunchKernelGGL((k_unpack_wgrad<float>), dim3(grid_for(total)), dim3(256), 0, st, dwp, O, I, k, Kpad, (float*)dw_oihw); break;
        case YOLO_BF16: hipLaunchKernelGGL((k_unpack_wgrad<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, dwp, O, I, k, Kpad, (bf16_t*)dw_oihw); break;
        case YOLO_F16:  hipLaunchKernelGGL((k_unpack_wgrad<f16_t>), dim3(grid_for(total)), dim3(256), 0, st, dwp, O, I, k, Kpad, (f16_t*)dw_oihw); break;
        default: return YOLO_ERR_DTYPE;
    }
    return YOLO_LAUNCH_CHECK();
}

// y[N,OH,OW,Cout] = conv(x[N,H,W,Cin], w) (+ bias); pad = k/2; wp = forward-packed weights in `dtype`.
// algo: 0 auto (MFMA when eligible), 1 generic VALU kernel, 2 MFMA or error.
// stats_acc (optional, needs bias == null): fp32 [8][2][Cout], pre-zeroed; receives sum(y) and sum(y^2) per channel
// (of the values as stored) for the BatchNorm that follows -- from the MFMA kernel's epilogue, no extra pass.
int yolo_conv2d_fwd(const void* x, int ldx, const void* wp, const float* bias, void* y, int ldy, float* stats_acc, int N,
                    int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, int algo,
                    hipStream_t st) {
    if (!supported(k, stride) || (stats_acc && bias)) return YOLO_ERR_ARG;
    int pad = k / 2;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return YOLO_ERR_ARG;
    ConvGeom g = fwd_geom(ldx, ldy, stats_acc, N, H, W, Cin, OH, OW, Cout, k, stride);
    return run_conv(g, x, wp, bias, y, 0, dtype, algo, st);
}

// Inference form of a fused Conv block (Model.fuse(): BatchNorm folded into the weights and a bias, reference
// src/model/model_blocks.py:36-37, src/utils/model_utils.py:72-118): y = act(conv(x) + bias) (+ res) in ONE launch -- bias,
// SiLU and the residual add of Residual / PSABlock ride in the epilogue of the MFMA kernels.  Returns 1 (nothing launched)
// when the shape / dtype has no MFMA kernel (fp32, unaligned channels): the caller then runs conv + the element-wise pass.
int yolo_conv2d_fwd_act(const void* x, int ldx, const void* wp, const float* bias, const void* res, int ldr, void* y, int ldy,
                        int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int act, int dtype,
                        hipStream_t st) {
    if (!supported(k, stride) || (act != 0 && act != 1)) return YOLO_ERR_ARG;
    int pad = k / 2;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return YOLO_ERR_ARG;
    if (res != nullptr && ((ldr & 3) || (reinterpret_cast<uintptr_t>(res) & 7))) return 1;
    ConvGeom g = fwd_geom(ldx, ldy, nullptr, N, H, W, Cin, OH, OW, Cout, k, stride);
    g.act = act; g.res = res; g.ldr = ldr;
    if (!mfma_conv_eligible(g, dtype, x, wp, y)) return 1;
    return mfma_conv_launch(g, x, wp, bias, y, 0, dtype, st);
}

// dx[N,H,W,Cin] (= or +=) conv^T(dy[N,OH,OW,Cout]); wb = dgrad-packed buffer from yolo_conv_pack_weights(mode 1)
static int conv2d_dgrad_impl(const void* dy, int lddy, const void* wb, void* dx, int lddx, const void* acc2, int ld2, int N, int H,
                            int W, int Cin, int OH, int OW, int Cout, int k, int stride, int accumulate, int dtype, int algo,
                            hipStream_t st);

int yolo_conv2d_dgrad(const void* dy, int lddy, const void* wb, void* dx, int lddx, int N, int H, int W, int Cin,
                      int OH, int OW, int Cout, int k, int stride, int accumulate, int dtype, int algo,
                      hipStream_t st) {
    return conv2d_dgrad_impl(dy, lddy, wb, dx, lddx, nullptr, 0, N, H, W, Cin, OH, OW, Cout, k, stride, accumulate, dtype, algo, st);
}

// dx = dgrad + dx + acc2: the data gradient accumulated into dx together with a SECOND tensor of dx's shape (row stride ld2)
// in the same epilogue -- a three-way gradient fan-in (C3K2: the chunk's half feeds the concat and a Residual whose own
// skip gradient is a third term) without an extra pass.  Stride 1 only.
int yolo_conv2d_dgrad_acc2(const void* dy, int lddy, const void* wb, void* dx, int lddx, const void* acc2, int ld2, int N, int H,
                           int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, int algo, hipStream_t st) {
    if (stride != 1 || acc2 == nullptr || (ld2 & 3) || (reinterpret_cast<uintptr_t>(acc2) & 7)) return YOLO_ERR_ARG;
    return conv2d_dgrad_impl(dy, lddy, wb, dx, lddx, acc2, ld2, N, H, W, Cin, OH, OW, Cout, k, stride, 1, dtype, algo, st);
}

static int conv2d_dgrad_impl(const void* dy, int lddy, const void* wb, void* dx, int lddx, const void* acc2, int ld2, int N, int H,
                            int W, int Cin, int OH, int OW, int Cout, int k, int stride, int accumulate, int dtype, int algo,
                            hipStream_t st) {
    if (!supported(k, stride)) return YOLO_ERR_ARG;
    size_t esz = dtype == YOLO_F32 ? 4 : 2;
    int ncls = stride == 2 ? 4 : 1;
    ConvGeom gs[4];
    long offs[4], off = 0;
    bool ring = algo != 1 && stride == 2;
    for (int c = 0; c < ncls; ++c) {
        gs[c] = dgrad_geom(lddy, lddx, N, H, W, Cin, OH, OW, Cout, k, stride, c);
        gs[c].acc2 = acc2; gs[c].ld2 = ld2;
        offs[c] = off;
        ring = ring && ring_conv_eligible(gs[c], dtype, dy, wb, dx);
        off += (long)Cin * gs[c].Kpad;
    }
    if (algo != 1 && stride == 2)                             // large maps: the dy patch once for all four classes (conv_up2.hip)
        if (const int uv = up2_conv_variant(gs, dtype)) return up2_conv_launch(gs, uv, offs, off, dy, wb, dx, accumulate, dtype, st);
    if (ring)                                                 // one launch for the four parity classes
        return ring_conv_launch(gs, 4, offs, off, dy, wb, nullptr, dx, accumulate, dtype, st);
    for (int c = 0; c < ncls; ++c) {
        const ConvGeom& g = gs[c];
        if (g.Hg > 0 && g.Wg > 0) {
            int rc = run_conv(g, dy, (const char*)wb + offs[c] * esz, nullptr, dx, accumulate, dtype, algo, st);
            if (rc) return rc;
        }
    }
    return YOLO_OK;
}

// Which kernel a forward / data-gradient launch takes (tests assert that their shapes reach the variant they mean to cover).
int yolo_conv2d_plan(int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int mode, int cls, int dtype) {
    if (!supported(k, stride)) return -1;
    const ConvGeom g = mode == 0 ? fwd_geom(Cin, Cout, nullptr, N, H, W, Cin, OH, OW, Cout, k, stride)
                                 : dgrad_geom(Cout, Cin, N, H, W, Cin, OH, OW, Cout, k, stride, cls);
    static const long long dummy[2] = {0, 0};               // eligibility looks at alignment only
    if (!mfma_conv_eligible(g, dtype, dummy, dummy, dummy)) return 0;
    if (mode == 1 && stride == 2) {                           // the four parity classes go out as one ring launch when all qualify
        ConvGeom gs[4];
        bool all = true;
        for (int c = 0; c < 4; ++c) {
            gs[c] = dgrad_geom(Cout, Cin, N, H, W, Cin, OH, OW, Cout, k, stride, c);
            all = all && ring_conv_eligible(gs[c], dtype, dummy, dummy, dummy);
        }
        if (const int uv = up2_conv_variant(gs, dtype)) return 5000 + uv;
        if (all) return ring_conv_plan(gs, 4);
    }
    return mfma_conv_plan(g, dtype);
}

long yolo_conv2d_wgrad_plan(int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype) {
    if (!supported(k, stride)) return -1;
    static const long long dummy[2] = {0, 0};
    if (!mfma_wgrad_eligible(Cin, Cout, Cin, Cout, dtype, dummy, dummy)) return 0;
    return mfma_wgrad2_plan(round_up32(k * k * Cin), N, H, W, Cin, OH, OW, Cout, k);
}

// Which weight-gradient kernel a call takes: 2 = MFMA, per-slab partial matrices + reduce (the product path),
// 3 = first MFMA design (atomics into one matrix; A/B runs and tensors >= 2^30 elements), 1 = generic.
static int wgrad_path(const void* x, int ldx, const void* dy, int ldy, int N, int H, int W, int Cin, int OH, int OW,
                      int Cout, int dtype, int algo) {
    if (algo != 1 && mfma_wgrad_eligible(Cin, Cout, ldx, ldy, dtype, x, dy)) {
        const bool big = (long)N * H * W * ldx + (long)(W + 1) * ldx >= (1L << 30) || (long)N * OH * OW * ldy >= (1L << 30);
        return (algo == 3 || big || (long)N * OH * OW == 0) ? 3 : 2;
    }
    return 1;
}

// fp32 scratch elements yolo_conv2d_wgrad needs for these arguments
long yolo_conv2d_wgrad_ws_elems(const void* x, int ldx, const void* dy, int ldy, int N, int H, int W, int Cin, int OH,
                                int OW, int Cout, int k, int stride, int dtype, int algo) {
    if (!supported(k, stride)) return 0;
    const int Kpad = round_up32(k * k * Cin);
    if (wgrad_path(x, ldx, dy, ldy, N, H, W, Cin, OH, OW, Cout, dtype, algo) == 2)
        return mfma_wgrad2_ws_elems(Kpad, N, H, W, Cin, OH, OW, Cout, k);
    return (long)Cout * Kpad;
}

// dw_oihw[Cout][Cin][k][k] (dw_dtype) = sum over pixels of dy (x) x.  ws: fp32 scratch of
// yolo_conv2d_wgrad_ws_elems(...) elements, contents undefined before and after (free for reuse on the stream).
int yolo_conv2d_wgrad(const void* x, int ldx, const void* dy, int ldy, float* ws, void* dw_oihw, int dw_dtype, int N,
                      int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, int algo,
                      hipStream_t st) {
    if (!supported(k, stride)) return YOLO_ERR_ARG;
    const int K = k * k * Cin, Kpad = round_up32(K);
    const int path = wgrad_path(x, ldx, dy, ldy, N, H, W, Cin, OH, OW, Cout, dtype, algo);
    if (path == 2)
        return mfma_wgrad2_launch(x, ldx, dy, ldy, ws, dw_oihw, dw_dtype, Kpad, N, H, W, Cin, OH, OW, Cout, k, stride, dtype, st);
    if (algo == 2) return YOLO_ERR_ARG;
    // single packed matrix accumulated with atomics, then unpacked
    float* dwp = ws;
    int rc = yolo_zero_async(dwp, (size_t)Cout * Kpad * sizeof(float), st);
    if (rc) return rc;
    if (path == 3) {
        rc = mfma_wgrad_launch(x, ldx, dy, ldy, dwp, Kpad, N, H, W, Cin, OH, OW, Cout, k, stride, dtype, st);
    } else if (dtype == YOLO_F32 && f32_wgrad_eligible(x, ldx, dy, ldy, Cin, Cout)) {
        rc = f32_wgrad_launch((const float*)x, ldx, (const float*)dy, ldy, dwp, Kpad, N, H, W, Cin, OH, OW, Cout, k, stride, st);
    } else {
        long nrows = (long)N * OH;
        long elems = (long)Cout * K;
        int gx = (int)((elems + 255) / 256);
        long want_slabs = 4096 / (gx > 0 ? gx : 1);
        if (want_slabs < 1) want_slabs = 1;
        if (want_slabs > nrows) want_slabs = nrows;
        if (want_slabs < 1) want_slabs = 1;
        int rps = (int)((nrows + want_slabs - 1) / want_slabs);
        if (rps < 1) rps = 1;
        int gy = (int)((nrows + rps - 1) / rps);
        if (gy > 0)
            YOLO_DISPATCH_T(dtype, hipLaunchKernelGGL((k_wgrad_generic<T>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, ldx,
                                                      (const T*)dy, ldy, dwp, Kpad, N, H, W, Cin, OH, OW, Cout, k, stride, rps));
        rc = YOLO_LAUNCH_CHECK();
    }
    if (rc) return rc;
    return yolo_conv_unpack_wgrad(dwp, Cout, Cin, k, dw_oihw, dw_dtype, st);
}

}  // extern "C"

// depthwise 3x3 stride 1 pad 1; w = fp32 [C][9]
static dim3 dw_grid(int nrows, int cv) {
    int tpr = cv < 256 ? cv : 256;
    return dim3((unsigned)(nrows < 4096 ? nrows : 4096), (unsigned)ceil_div(cv, tpr));
}

// dwconv.hip: the strip kernels for 16-bit tensors with 8-channel alignment (-1 = does not qualify)
int dw_strip_launch(bool flip, const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C, int dtype,
                    int accumulate, float* stats, hipStream_t st, const float* bias = nullptr, int act = 0);
int dw_strip_wgrad_launch(const void* x, int ldx, const void* dy, int ldy, float* partial, int nslab, int N, int H, int W, int C,
                          int dtype, hipStream_t st);

template <bool FLIP>
static int dw_launch(const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C, int dtype,
                     int accumulate, hipStream_t st) {
    const int fast = dw_strip_launch(FLIP, x, ldx, w, y, ldy, N, H, W, C, dtype, accumulate, nullptr, st);
    if (fast >= 0) return fast;
    YOLO_DISPATCH_T(dtype, {
        if (vecok<T>(x, ldx, C) && vecok<T>(y, ldy, C)) {
            constexpr int V = vec_of<T>::N;
            hipLaunchKernelGGL((k_dw3x3<T, V, FLIP>), dw_grid(N * H, C / V), dim3(256), 0, st, (const T*)x, ldx, w, (T*)y,
                               ldy, N, H, W, C, accumulate);
        } else {
            hipLaunchKernelGGL((k_dw3x3<T, 1, FLIP>), dw_grid(N * H, C), dim3(256), 0, st, (const T*)x, ldx, w, (T*)y, ldy,
                               N, H, W, C, accumulate);
        }
    });
    return YOLO_LAUNCH_CHECK();
}

extern "C" {

int yolo_dwconv3x3_fwd(const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C, int dtype,
                       hipStream_t st) {
    return dw_launch<false>(x, ldx, w, y, ldy, N, H, W, C, dtype, 0, st);
}

// forward that also accumulates the BatchNorm batch statistics of y (sum, sum of squares of the stored values) into
// stats[8][2][C] (zeroed by the caller; the layout yolo_bn_stats_acc fills).  Returns 1 when the fused kernel does not take
// these tensors (fp32, unaligned): nothing was launched, the caller runs yolo_dwconv3x3_fwd + yolo_bn_stats_acc.
int yolo_dwconv3x3_fwd_stats(const void* x, int ldx, const float* w, void* y, int ldy, float* stats, int N, int H, int W, int C,
                             int dtype, hipStream_t st) {
    const int fast = dw_strip_launch(false, x, ldx, w, y, ldy, N, H, W, C, dtype, 0, stats, st);
    return fast < 0 ? 1 : fast;
}

// y = act(dwconv(x) + bias) in one launch: the depthwise blocks of a fused model (Model.fuse(): BatchNorm folded into w and
// bias).  act: 0 identity, 1 SiLU.  Returns 1 when the strip kernel does not take these tensors (nothing was launched; the
// caller runs yolo_dwconv3x3_fwd + the element-wise pass).
int yolo_dwconv3x3_fwd_act(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int H, int W, int C,
                           int act, int dtype, hipStream_t st) {
    if (bias == nullptr || (act != 0 && act != 1)) return YOLO_ERR_ARG;
    const int fast = dw_strip_launch(false, x, ldx, w, y, ldy, N, H, W, C, dtype, 0, nullptr, st, bias, act);
    return fast < 0 ? 1 : fast;
}

int yolo_dwconv3x3_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int H, int W, int C,
                         int accumulate, int dtype, hipStream_t st) {
    return dw_launch<true>(dy, lddy, w, dx, lddx, N, H, W, C, dtype, accumulate, st);
}

// number of row slabs (= partial blocks) yolo_dwconv3x3_wgrad uses for an (N, H) map
int yolo_dw_wgrad_nslab(int N, int H) {
    int nrows = N * H;
    int slabs = nrows < 1024 ? nrows : 1024;
    int rps = (nrows + slabs - 1) / slabs;
    return (nrows + rps - 1) / rps;
}

// dw fp32 [C][9]; partial: fp32 scratch [nslab][C][9]
int yolo_dwconv3x3_wgrad(const void* x, int ldx, const void* dy, int ldy, float* dw, float* partial, int N, int H,
                         int W, int C, int dtype, hipStream_t st) {
    int nrows = N * H;
    int slabs = yolo_dw_wgrad_nslab(N, H);
    int rps = (nrows + slabs - 1) / slabs;
    const int fast = dw_strip_wgrad_launch(x, ldx, dy, ldy, partial, slabs, N, H, W, C, dtype, st);
    if (fast > 0) return fast;
    if (fast < 0) YOLO_DISPATCH_T(dtype, {
        if (vecok<T>(x, ldx, C) && vecok<T>(dy, ldy, C)) {
            constexpr int V = vec_of<T>::N;
            int cv = C / V, tpr = cv < 256 ? cv : 256;
            hipLaunchKernelGGL((k_dw3x3_wgrad<T, V>), dim3(slabs, ceil_div(cv, tpr)), dim3(256), 0, st, (const T*)x, ldx,
                               (const T*)dy, ldy, partial, N, H, W, C, rps);
        } else {
            int tpr = C < 256 ? C : 256;
            hipLaunchKernelGGL((k_dw3x3_wgrad<T, 1>), dim3(slabs, ceil_div(C, tpr)), dim3(256), 0, st, (const T*)x, ldx,
                               (const T*)dy, ldy, partial, N, H, W, C, rps);
        }
    });
    hipLaunchKernelGGL(k_dw_wgrad_finalize, dim3(ceil_div(C * 9, 32)), dim3(1024), 0, st, partial, slabs, C * 9, dw);
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
