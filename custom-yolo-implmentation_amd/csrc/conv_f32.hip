// fp32 convolution (forward, data gradient, weight gradient) as LDS-tiled implicit GEMMs on the vector ALUs.
// fp32 is the PARITY path (the goldens produced by the reference are fp32; the training metric runs in bf16 on the
// MFMA kernels): the first version of these kernels computed one output element per thread straight from global
// memory (2.3 TFLOP/s on preset x).  Here a 256-thread workgroup owns a 64 x 64 output tile, stages 16-deep K slices
// of both operands in LDS (k-major, so a thread's four rows / columns are one 16-byte LDS read) and every thread
// keeps a 4 x 4 register block: 16 FMAs per 8 LDS floats.  Same geometry contract as the other conv kernels
// (conv_geom.h); summation order differs from the element-wise kernel only in the grouping of the K loop.
#include "common.h"
#include "conv_geom.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

// dst[pixel][cd] (+)= bias[cd] + sum_{tap, cs} src[pixel (+) tap][cs] * wm[cd][tap*Cs + cs]
template <bool ACC>
__global__ __launch_bounds__(256) void k_conv_f32(ConvGeom g, const float* __restrict__ src, const float* __restrict__ wm,
                                                  const float* __restrict__ bias, float* __restrict__ dst, int ntile_n) {
    __shared__ __attribute__((aligned(16))) float As[TK][TM + 4];     // [k][pixel]
    __shared__ __attribute__((aligned(16))) float Bs[TK][TN + 4];     // [k][channel]
    const int tid = threadIdx.x;
    const int tile_m = blockIdx.x / ntile_n, tile_n = blockIdx.x - tile_m * ntile_n;
    const long m0 = (long)tile_m * TM;
    const int cd0 = tile_n * TN;
    const long total_pix = (long)g.N * g.Hg * g.Wg;
    // loader role: row lr (a pixel of the A tile, a channel of the B tile), K quad kq
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    const long q = m0 + lr;
    const bool pv = q < total_pix;
    const long qq = pv ? q : 0;
    const int lb = (int)(qq % g.Wg);
    const long t2 = qq / g.Wg;
    const int la = (int)(t2 % g.Hg);
    const long ln = t2 / g.Hg;
    const bool cv = cd0 + lr < g.Cd;
    const float* wrow = wm + (long)(cv ? cd0 + lr : 0) * g.Kpad;
    // compute role: pixels ty*4.., channels tx*4..
    const int tx = tid & 15, ty = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int t = 0; t < g.ntaps; ++t) {
        const int hs = la * g.sstride + g.dh[t], ws = lb * g.sstride + g.dw[t];
        const bool tv = pv && hs >= 0 && hs < g.Hs && ws >= 0 && ws < g.Ws;
        const float* sp = src + ((ln * g.Hs + (tv ? hs : 0)) * g.Ws + (tv ? ws : 0)) * (long)g.lds;
        const float* wp = wrow + (long)t * g.Cs;
        for (int c0 = 0; c0 < g.Cs; c0 += TK) {
            float4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
            const int c = c0 + kq;
            if (c < g.Cs) {                                  // Cs % 4 == 0: a quad is all-in or all-out
                if (tv) av = *reinterpret_cast<const float4*>(sp + c);
                if (cv) bv = *reinterpret_cast<const float4*>(wp + c);
            }
            __syncthreads();                                 // the previous slice has been consumed
            As[kq + 0][lr] = av.x; As[kq + 1][lr] = av.y; As[kq + 2][lr] = av.z; As[kq + 3][lr] = av.w;
            Bs[kq + 0][lr] = bv.x; Bs[kq + 1][lr] = bv.y; Bs[kq + 2][lr] = bv.z; Bs[kq + 3][lr] = bv.w;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) {
                const float4 a = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
                const float4 b = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
                const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
            }
        }
    }
    const int c = cd0 + tx * 4;
    if (c >= g.Cd) return;                                   // Cd % 4 == 0
    float bvv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr)
#pragma unroll
        for (int j = 0; j < 4; ++j) bvv[j] = bias[c + j];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long p = m0 + ty * 4 + i;
        if (p >= total_pix) break;
        const int b = (int)(p % g.Wg);
        const long t3 = p / g.Wg;
        const int a = (int)(t3 % g.Hg);
        const long n = t3 / g.Hg;
        float* d = dst + ((n * g.Hd + a * g.ostep + g.ooff_h) * g.Wd + b * g.ostep + g.ooff_w) * (long)g.ldd + c;
        float4 v = {acc[i][0] + bvv[0], acc[i][1] + bvv[1], acc[i][2] + bvv[2], acc[i][3] + bvv[3]};
        if (ACC) {
            const float4 o = *reinterpret_cast<const float4*>(d);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(d) = v;
    }
}

// dwp[co][tap*Cin + ci] += sum over the output rows [r0, r1) of dy[p][co] * x[p (+) tap][ci]   (atomics across slabs)
__global__ __launch_bounds__(256) void k_wgrad_f32(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy,
                                                   float* __restrict__ dwp, int Kpad, int N, int H, int W, int Cin, int OH,
                                                   int OW, int Cout, int k, int stride, int rows_per_slab, int ntile_k) {
    __shared__ __attribute__((aligned(16))) float As[TK][TM + 4];     // [pixel][co]
    __shared__ __attribute__((aligned(16))) float Bs[TK][TN + 4];     // [pixel][k column]
    const int tid = threadIdx.x;
    const int tile_co = blockIdx.x / ntile_k, tile_k = blockIdx.x - tile_co * ntile_k;
    const int co0 = tile_co * TM, k0 = tile_k * TN;
    const int K = k * k * Cin, pad = k / 2;
    const long nrows = (long)N * OH;
    const long r0 = (long)blockIdx.y * rows_per_slab;
    long r1 = r0 + rows_per_slab;
    if (r1 > nrows) r1 = nrows;
    const long p0 = r0 * OW, p1 = r1 * OW;
    // loader role: pixel lp of the slice, quad lq of the 64 columns
    const int lp = tid >> 4, lq = (tid & 15) * 4;
    const int kc = k0 + lq;                                  // this thread's k column quad (Cin % 4 == 0: one tap)
    const bool kv = kc < K;
    const int tap = kv ? kc / Cin : 0, ci = kv ? kc - tap * Cin : 0;
    const int kh = tap / k, kw = tap - kh * k;
    const bool av = co0 + lq < Cout;
    const int tx = tid & 15, ty = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (long pb = p0; pb < p1; pb += TK) {
        const long p = pb + lp;
        float4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
        if (p < p1) {
            const int ow = (int)(p % OW);
            const long r = p / OW;
            const int oh = (int)(r % OH);
            const long n = r / OH;
            if (av) a4 = *reinterpret_cast<const float4*>(dy + p * (long)ldy + co0 + lq);
            const int ih = oh * stride + kh - pad, iw = ow * stride + kw - pad;
            if (kv && ih >= 0 && ih < H && iw >= 0 && iw < W)
                b4 = *reinterpret_cast<const float4*>(x + ((n * H + ih) * W + iw) * (long)ldx + ci);
        }
        __syncthreads();
        *reinterpret_cast<float4*>(&As[lp][lq]) = a4;
        *reinterpret_cast<float4*>(&Bs[lp][lq]) = b4;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            const float4 a = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
            const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty * 4 + i;
        if (co >= Cout) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kcol = k0 + tx * 4 + j;
            if (kcol < K) atomicAdd(dwp + (long)co * Kpad + kcol, acc[i][j]);
        }
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// 1 if the tiled fp32 kernel covers this launch (16-byte loads and stores everywhere)
int f32_conv_eligible(const ConvGeom& g, const void* src, const void* wm, const void* dst) {
    return g.Cs % 4 == 0 && g.lds % 4 == 0 && g.Cd % 4 == 0 && g.ldd % 4 == 0 && g.Kpad % 4 == 0 && al16(src) && al16(wm) &&
           al16(dst) && (long)g.N * g.Hg * g.Wg > 0;
}

int f32_conv_launch(const ConvGeom& g, const float* src, const float* wm, const float* bias, float* dst, int accumulate,
                    hipStream_t st) {
    const long pix = (long)g.N * g.Hg * g.Wg;
    const long tm = (pix + TM - 1) / TM;
    const int tn = (g.Cd + TN - 1) / TN;
    if (tm * tn > 0x7fffffffL) return YOLO_ERR_ARG;
    if (accumulate) hipLaunchKernelGGL((k_conv_f32<true>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, g, src, wm, bias, dst, tn);
    else hipLaunchKernelGGL((k_conv_f32<false>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, g, src, wm, bias, dst, tn);
    return YOLO_LAUNCH_CHECK();
}

int f32_wgrad_eligible(const void* x, int ldx, const void* dy, int ldy, int Cin, int Cout) {
    return Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && al16(x) && al16(dy);
}

// dwp: zeroed fp32 [Cout][Kpad] packed matrix (k column = tap*Cin + ci)
int f32_wgrad_launch(const float* x, int ldx, const float* dy, int ldy, float* dwp, int Kpad, int N, int H, int W, int Cin,
                     int OH, int OW, int Cout, int k, int stride, hipStream_t st) {
    const int K = k * k * Cin;
    const int tco = (Cout + TM - 1) / TM, tk = (K + TN - 1) / TN;
    const long nrows = (long)N * OH;
    if (nrows == 0) return YOLO_OK;
    long want_slabs = 2048 / ((long)tco * tk);               // ~8 workgroups per CU in total
    if (want_slabs < 1) want_slabs = 1;
    if (want_slabs > nrows) want_slabs = nrows;
    const int rps = (int)((nrows + want_slabs - 1) / want_slabs);
    const int gy = (int)((nrows + rps - 1) / rps);
    hipLaunchKernelGGL(k_wgrad_f32, dim3((unsigned)(tco * tk), (unsigned)gy), dim3(256), 0, st, x, ldx, dy, ldy, dwp, Kpad, N,
                       H, W, Cin, OH, OW, Cout, k, stride, rps, tk);
    return YOLO_LAUNCH_CHECK();
}
