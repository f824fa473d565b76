// Geometry of one "gather convolution" launch, shared by the generic and the MFMA kernels.
//
//   dst[n, a*ostep+ooff_h, b*ostep+ooff_w, cd] (+)= bias[cd] +
//        sum_t sum_cs src[n, a*sstride+dh[t], b*sstride+dw[t], cs] * Wm[cd][t*Cs + cs]
//
// for (a, b) over the Hg x Wg destination grid; source taps outside [0,Hs)x[0,Ws) contribute 0.
//   forward k x k stride s pad k/2 : ostep 1, sstride s, dh = kh - pad, src = x,  Wm = w[co][kh][kw][ci]
//   dgrad stride 1                  : same with src = dy and Wm = w[ci][k-1-kh'][k-1-kw'][co]
//   dgrad stride 2 (k 3, pad 1)     : four parity classes (ph,pw) of dx pixels, each with only its
//                                     valid taps (1, 2, 2, 4 of the 9): ostep 2, sstride 1
#pragma once
#include <stdint.h>

struct ConvGeom {
    int N, Hs, Ws, Cs, lds;
    int Hd, Wd, Cd, ldd;
    int Hg, Wg;
    int ostep, ooff_h, ooff_w, sstride;
    int ntaps;
    int dh[9], dw[9];
    int K, Kpad;
    float* stats;   // forward only: [8][2][Cd] fp32 accumulator for BatchNorm batch statistics, or null
    const void* acc2;   // accumulating launches only: a SECOND tensor (same pixels and channels as dst, row stride ld2) added
    int ld2;            // to the result as well -- dst = conv + dst + acc2 (C3K2: the chunk's gradient fan-in), or null
    // inference epilogue (Model.fuse(): BatchNorm folded into weights + bias, model_blocks.py:36-37): dst = act(conv + bias) + res
    int act;            // YOLO_ACT_IDENTITY / YOLO_ACT_SILU, applied after the bias
    const void* res;    // optional residual (same pixels / channels as dst, row stride ldr) added after the activation, or null
    int ldr;
};

static inline int round_up32(int k) { return (k + 31) / 32 * 32; }

// taps of dgrad class `cls` (= ph*2+pw) for k=3, stride 2, pad 1: flipped index kh' with
// (ih - 1 + kh') even; dy row = a + (ph - 1 + kh')/2 where ih = 2a + ph.
static inline int dgrad_s2_taps_1d(int ph, int* kflip, int* d) {
    if (ph == 0) { kflip[0] = 1; d[0] = 0; return 1; }
    kflip[0] = 0; d[0] = 0; kflip[1] = 2; d[1] = 1; return 2;
}

// Fills taps for mode 0 (forward) / 1 (dgrad).  khs/kws receive, per tap, the ORIGINAL kernel
// indices (kh, kw) of w[co][ci][kh][kw] that multiply that tap (used by the weight packers).
static inline int conv_taps(int mode, int k, int stride, int cls, int* dh, int* dw, int* khs, int* kws) {
    int pad = k / 2, n = 0;
    if (mode == 0 || stride == 1) {
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) {
                dh[n] = a - pad; dw[n] = b - pad;
                khs[n] = mode == 0 ? a : k - 1 - a;
                kws[n] = mode == 0 ? b : k - 1 - b;
                ++n;
            }
        return n;
    }
    int fh[2], dhh[2], fw[2], dww[2];
    int nh = dgrad_s2_taps_1d(cls >> 1, fh, dhh), nw = dgrad_s2_taps_1d(cls & 1, fw, dww);
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
            dh[n] = dhh[a]; dw[n] = dww[b];
            khs[n] = 2 - fh[a]; kws[n] = 2 - fw[b];
            ++n;
        }
    return n;
}

// Kernel-selection overrides (tile widths, K order, halo variant, LDS-DMA, ring kernel on/off, ring depth); 0 / -1 =
// automatic.  Read ONCE per process from YOLO_CONV_TUNE ("bn,tap_inner,halo,dma,ring,bm,nst,bk"); tools/conv_tune.py and the
// variant-forcing parity tests change them through yolo_conv_tune_set.  Nothing on the training path writes them.
struct ConvTune { int bn, tap_inner, halo, dma, ring, bm, nst, bk; };
ConvTune& conv_tune();
