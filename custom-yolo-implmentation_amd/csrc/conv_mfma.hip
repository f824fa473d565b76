// bf16 / f16 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_{bf16,f16}).
//
// Forward and data-gradient share one kernel (ConvGeom, conv_geom.h): a 128-pixel x BN-channel
// output tile per 256-thread workgroup (4 waves), K = taps*Cs walked in 32-element steps.  The
// activation tile is gathered NHWC row by row (16 B = 8 channels per lane, zero for padding taps),
// the weight tile comes from the K-major packed matrix; both are staged through LDS (80-byte
// rows: 64 B payload + 16 B pad to spread ds_read_b128 over the banks) with a register-staged
// double buffer, one barrier per K-step.  MFMA roles are swapped (A = weights, B = activations) so a
// lane's four accumulator registers are four CONSECUTIVE output channels of one pixel: the epilogue
// stores 8 bytes per lane into NHWC rows.
//
// Weight-gradient: dW[co][tap][ci] = sum_p dY[p][co] * X[p+tap][ci].  The reduction index is the
// pixel, which is the slow axis of NHWC, so both tiles are transposed on their way into LDS
// (pixel-major inner) and the K loop runs over 32-pixel steps of one slab; slabs are combined with
// fp32 atomics into the packed gradient matrix.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "conv_geom.h"
#include "conv_dev.h"

int halo_conv_eligible(const ConvGeom& g);
int rows_conv_eligible(const ConvGeom& g);
int rows_conv_launch(const ConvGeom& g, int variant, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                     int dtype, hipStream_t st);
int halo_conv_launch(const ConvGeom& g, int variant, const void* src, const void* wm, const float* bias, void* dst,
                     int accumulate, int dtype, hipStream_t st);

ConvTune& conv_tune() {
    static ConvTune t = [] {
        ConvTune v{0, -1, -1, -1, -1, 0, 0, 0};
        if (const char* e = getenv("YOLO_CONV_TUNE"))
            sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &v.bn, &v.tap_inner, &v.halo, &v.dma, &v.ring, &v.bm, &v.nst, &v.bk);
        return v;
    }();
    return t;
}

int ring_conv_eligible(const ConvGeom& g, int dtype, const void* src, const void* wm, const void* dst);
int ring_conv_plan(const ConvGeom* gs, int n);
int ring_conv_launch(const ConvGeom* gs, int n, const long* wm_off, long wm_elems, const void* src, const void* wm,
                     const float* bias, void* dst, int accumulate, int dtype, hipStream_t st);

namespace {

// MODE 0: K walked tap-major (k = tap*Cs + ch), any Cs % 8 == 0; each lane tracks its own (tap, ch).
// MODE 1: the same with Cs < 32 (a step can cross several taps).
// MODE 2: Cs % 32 == 0.  K is walked channel-chunk-major with the taps innermost (a workgroup re-reads its
//   input patch for all taps of one 32-channel chunk back to back, so the re-reads hit L2), the tap of a step is
//   wave-uniform, and both operands come through buffer descriptors: the per-lane byte offset (voffset) is fixed
//   for the whole loop, the step's tap / chunk / weight-column offset is one scalar (soffset), and a padding tap
//   or a row past the end is a voffset beyond the descriptor's range, which the hardware reads as zero -- the
//   gather costs 3 vector instructions per row and step.  All sizes are below 2^30 elements (host check).
// MODE 3: MODE 2 with LDS-DMA (buffer_load_dwordx4 ... lds): the tiles go global -> LDS without passing through
//   registers (no ds_write pass, 16 fewer VGPRs).  A wave instruction writes 64 x 16 B contiguously, which is exactly
//   16 rows of the swizzled 64-byte-row image when the lane in slot s of a row fetches chunk s ^ swizzle(row);
//   out-of-range lanes write zeros (tests/test_gpu_selftest.py pins both facts).
template <typename T, int WGM, int WGN, int WM, int WN, bool ACC, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_conv_mfma(GeomDev g, const T* __restrict__ src, const T* __restrict__ wm,
                                                   const float* __restrict__ bias, T* __restrict__ dst, int ntile_n) {
    static_assert(WGM * WGN == 4 && WGM * WM * 16 == BM, "tile shape");
    constexpr int BN = WGN * WN * 16;
    using ops = mfma_ops<T>;
    using frag = typename ops::frag;
    __shared__ __attribute__((aligned(16))) T lds_a[2][BM][LDSROW];
    __shared__ __attribute__((aligned(16))) T lds_b[2][BN][LDSROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tile_m = tile / ntile_n, tile_n = tile - tile_m * ntile_n;
    const int m0 = tile_m * BM;
    const int cd0 = tile_n * BN;
    const int total_pix = g.N * g.Hg * g.Wg;

    // ---- loader state: two activation rows (r, r+64) and up to two weight rows per thread, fixed k-segment
    const int kseg = tid & 3;
    const int lrow = tid >> 2;                      // 0..63
    const int sk = (kseg ^ ((-(lrow >> 2)) & 3)) * 8;   // swizzled chunk this thread stores
    int rowoff[2], hs0[2], ws0[2];                  // element offset of the row's (n, hs0, ws0) source pixel
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = m0 + lrow + i * 64;
        const bool rv = q < total_pix;
        const unsigned qq = rv ? q : 0;
        const unsigned t2 = qq / (unsigned)g.Wg, b = qq - t2 * g.Wg;
        const unsigned n = t2 / (unsigned)g.Hg, a = t2 - n * g.Hg;
        hs0[i] = rv ? (int)a * g.sstride : -(1 << 20);     // rows past the end fail every bounds test
        ws0[i] = (int)b * g.sstride;
        rowoff[i] = (((int)n * g.Hs + (int)a * g.sstride) * g.Ws + (int)b * g.sstride) * g.lds;
    }
    constexpr int WR = (BN + 63) / 64;              // weight rows per thread (lrow, lrow+64)
    bool wvalid[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int r = lrow + i * 64;
        wvalid[i] = (r < BN) && (cd0 + r < g.Cd);
    }

    uint4 ra[2], rb[WR];

    // MODE 2 state
    unsigned vmask[2] = {0u, 0u};                   // bit t: tap t of this row is inside the source image
    int voffa[2], voffb[WR];
    int tap = 0, cbase = 0, wcol = 0;               // uniform
    __amdgpu_buffer_rsrc_t rsa, rsb;
    // MODE 0/1 state
    int ltap = 0, ch = kseg * 8;
    const T* wrow[WR];

    if constexpr (MODE >= 2) {
        for (int t = 0; t < g.ntaps; ++t) {
            const int dh = (int)((g.dh_pack >> (2 * t)) & 3u) - 1, dw = (int)((g.dw_pack >> (2 * t)) & 3u) - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = (unsigned)(hs0[i] + dh) < (unsigned)g.Hs && (unsigned)(ws0[i] + dw) < (unsigned)g.Ws;
                vmask[i] |= (ok ? 1u : 0u) << t;
            }
        }
        // descriptor base = src - one row - one pixel, so the scalar tap offset (dh+1, dw+1) is never negative
        const int shift = (g.Ws + 1) * g.lds;
        rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src) - shift, 0, (g.N * g.Hs * g.Ws * g.lds + shift) * 2,
                                                0x00020000);
        rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wm), 0, g.Cd * g.Kpad * 2, 0x00020000);
        // register staging: this thread loads chunk kseg and stores it at the swizzled slot; LDS-DMA: the lane's
        // slot is fixed (lane-linear image), so it loads the chunk that belongs there
        const int kload = MODE == 3 ? (kseg ^ ((-(lrow >> 2)) & 3)) : kseg;
#pragma unroll
        for (int i = 0; i < 2; ++i) voffa[i] = (rowoff[i] + kload * 8) * 2;
#pragma unroll
        for (int i = 0; i < WR; ++i)
            voffb[i] = wvalid[i] ? ((cd0 + lrow + i * 64) * g.Kpad + kload * 8) * 2 : (int)0x80000000;
    } else {
#pragma unroll
        for (int i = 0; i < WR; ++i) wrow[i] = wm + (long)(cd0 + (wvalid[i] ? lrow + i * 64 : 0)) * g.Kpad + kseg * 8;
        while (ch >= g.Cs) { ch -= g.Cs; ++ltap; }
    }

    auto gload = [&](int kt) {
        if constexpr (MODE >= 2) {
            const int soff = ((int)((g.dh_pack >> (2 * tap)) & 3u) * g.Ws + (int)((g.dw_pack >> (2 * tap)) & 3u)) * g.lds + cbase;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int vo = ((vmask[i] >> tap) & 1u) ? voffa[i] : (int)0x80000000;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsa, vo, soff * 2, 0);
                ra[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsb, voffb[i], wcol * 2, 0);
                rb[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
            if (g.tap_inner) {
                ++tap;
                wcol += g.Cs;
                if (tap == g.ntaps) { tap = 0; cbase += BK; wcol += BK - g.ntaps * g.Cs; }
            } else {
                cbase += BK;
                wcol += BK;
                if (cbase == g.Cs) { cbase = 0; ++tap; }
            }
        } else {
            const bool tv = ltap < g.ntaps;
            const int dh = (int)((g.dh_pack >> (2 * ltap)) & 3u) - 1, dw = (int)((g.dw_pack >> (2 * ltap)) & 3u) - 1;
            const int tapoff = (dh * g.Ws + dw) * g.lds + ch;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = tv && (unsigned)(hs0[i] + dh) < (unsigned)g.Hs && (unsigned)(ws0[i] + dw) < (unsigned)g.Ws;
                const uint4 v = *reinterpret_cast<const uint4*>(src + (ok ? rowoff[i] + tapoff : 0));
                ra[i] = ok ? v : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                const uint4 v = *reinterpret_cast<const uint4*>(wrow[i] + kt * BK);
                rb[i] = wvalid[i] ? v : make_uint4(0, 0, 0, 0);
            }
            ch += BK;
            if (MODE == 1) { while (ch >= g.Cs) { ch -= g.Cs; ++ltap; } }
            else if (ch >= g.Cs) { ch -= g.Cs; ++ltap; }
        }
    };
    // MODE 3: the same step as one LDS-DMA instruction per 16-row group (wave w: rows 16w.. and 64+16w..)
    auto gload_dma = [&](int buf) {
        const int soff = ((int)((g.dh_pack >> (2 * tap)) & 3u) * g.Ws + (int)((g.dw_pack >> (2 * tap)) & 3u)) * g.lds + cbase;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int vo = ((vmask[i] >> tap) & 1u) ? voffa[i] : (int)0x80000000;
            lds_dma16(rsa, &lds_a[buf][wave * 16 + i * 64][0], vo, soff * 2);
        }
#pragma unroll
        for (int i = 0; i < WR; ++i)
            if (BN >= (i + 1) * 64 || wave * 16 + i * 64 < BN)
                lds_dma16(rsb, &lds_b[buf][wave * 16 + i * 64][0], voffb[i], wcol * 2);
        if (g.tap_inner) {
            ++tap;
            wcol += g.Cs;
            if (tap == g.ntaps) { tap = 0; cbase += BK; wcol += BK - g.ntaps * g.Cs; }
        } else {
            cbase += BK;
            wcol += BK;
            if (cbase == g.Cs) { cbase = 0; ++tap; }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<uint4*>(&lds_a[buf][lrow + i * 64][sk]) = ra[i];
#pragma unroll
        for (int i = 0; i < WR; ++i)
            if (BN >= (i + 1) * 64 || lrow + i * 64 < BN) *reinterpret_cast<uint4*>(&lds_b[buf][lrow + i * 64][sk]) = rb[i];
    };

    // ---- compute state
    const int wgm = wave / WGN, wgn = wave - wgm * WGN;
    const int prow = wgm * WM * 16, crow = wgn * WN * 16;
    const int fr = lane & 15;
    const int fk = ((lane >> 4) ^ ((-(fr >> 2)) & 3)) * 8;     // swizzled chunk of this lane's fragment rows
    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (MODE == 3) {
        gload_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < g.KT; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < g.KT) gload_dma(buf ^ 1);      // the other stage was last read before the previous barrier
            frag fa[WN], fb[WM];
#pragma unroll
            for (int j = 0; j < WN; ++j) fa[j] = *reinterpret_cast<const frag*>(&lds_b[buf][crow + j * 16 + fr][fk]);
#pragma unroll
            for (int i = 0; i < WM; ++i) fb[i] = *reinterpret_cast<const frag*>(&lds_a[buf][prow + i * 16 + fr][fk]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = ops::mma(fa[j], fb[i], acc[i][j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        gload(0);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < g.KT; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < g.KT) gload(kt + 1);
            frag fa[WN], fb[WM];
#pragma unroll
            for (int j = 0; j < WN; ++j) fa[j] = *reinterpret_cast<const frag*>(&lds_b[buf][crow + j * 16 + fr][fk]);
#pragma unroll
            for (int i = 0; i < WM; ++i) fb[i] = *reinterpret_cast<const frag*>(&lds_a[buf][prow + i * 16 + fr][fk]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = ops::mma(fa[j], fb[i], acc[i][j]);
            if (kt + 1 < g.KT) lstore(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds channels cbase..cbase+3 of pixel (tile pixel i*16 + fr).  The first pixel's
    // coordinates come from two divisions, the following ones (+16 pixels each) by carrying.
    const int cq = (lane >> 4) * 4;
    float bv[WN][4];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int c = cd0 + crow + j * 16 + cq;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = (bias != nullptr && c < g.Cd) ? bias[c + r] : 0.f;
    }
    {
        int q = m0 + prow + fr;
        const unsigned qq = q < total_pix ? q : 0;
        const unsigned t2 = qq / (unsigned)g.Wg;
        int b = (int)(qq - t2 * g.Wg);
        int n = (int)(t2 / (unsigned)g.Hg);
        int a = (int)t2 - n * g.Hg;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const bool live = q < total_pix;
            const long pix = live ? ((long)n * g.Hd + a * g.ostep + g.ooff_h) * (long)g.Wd + b * g.ostep + g.ooff_w : 0;
            store_pixel_blocks<T, WN, ACC>(g, acc[i], bv, dst, pix, live, cd0 + crow, cq, lane);
            q += 16;
            b += 16;
            while (b >= g.Wg) {
                b -= g.Wg;
                if (++a == g.Hg) { a = 0; ++n; }
            }
        }
    }

    // ---- optional: per-channel sum / sum of squares of the values just stored (rounded to T), added to
    // replica (workgroup mod 8) of stats[8][2][Cd]: BatchNorm batch statistics without a second pass over y
    float* const stats = g.stats;
    if (stats != nullptr) {
        float* sacc = reinterpret_cast<float*>(&lds_a[0][0][0]);          // [2][BN]; LDS is idle after the K loop
        for (int t = tid; t < 2 * BN; t += 256) sacc[t] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float s[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = to_f<T>(from_f<T>(acc[i][j][r]));     // rows past the last pixel hold 0
                    s[r] += v;
                    q2[r] += v * v;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r] = row16_sum(s[r]);
                q2[r] = row16_sum(q2[r]);
            }
            if (fr == 0) {
                const int cl = crow + j * 16 + cq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atomicAdd(&sacc[cl + r], s[r]);
                    atomicAdd(&sacc[BN + cl + r], q2[r]);
                }
            }
        }
        __syncthreads();
        float* o = stats + (long)(blockIdx.x & 7) * 2 * g.Cd;
        for (int t = tid; t < BN; t += 256)
            if (cd0 + t < g.Cd) {
                atomicAdd(o + cd0 + t, sacc[t]);
                atomicAdd(o + g.Cd + cd0 + t, sacc[BN + t]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
constexpr int WG_BP = 32;   // pixels per K-step
constexpr int WG_LDSROW = 40;   // 32 + 8 pad elements per LDS row

template <typename T>
__global__ __launch_bounds__(256) void k_wgrad_mfma(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int ldy,
                                                    float* __restrict__ dwp, int Kpad, int N, int H, int W, int Cin,
                                                    int OH, int OW, int Cout, int k, int stride, int ci_tiles,
                                                    long slab_pix) {
    using ops = mfma_ops<T>;
    using frag = typename ops::frag;
    __shared__ __attribute__((aligned(16))) T lds_y[64][WG_LDSROW];   // [co][pixel]
    __shared__ __attribute__((aligned(16))) T lds_x[64][WG_LDSROW];   // [ci][pixel]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * 64;
    const int tapi = blockIdx.y / ci_tiles;
    const int ci0 = (blockIdx.y - tapi * ci_tiles) * 64;
    const int kh = tapi / k, kw = tapi - kh * k, pad = k / 2;
    const long P = (long)N * OH * OW;
    const long q0 = (long)blockIdx.z * slab_pix;
    long q1 = q0 + slab_pix;
    if (q1 > P) q1 = P;
    const int steps = (int)((q1 - q0 + WG_BP - 1) / WG_BP);

    const int pix = tid & 31, seg = tid >> 5;       // 32 pixels x 8 channel segments
    const bool yv = co0 + seg * 8 < Cout, xv = ci0 + seg * 8 < Cin;

    uint4 ry, rx;
    auto gload = [&](int s) {
        long q = q0 + (long)s * WG_BP + pix;
        ry = make_uint4(0, 0, 0, 0);
        rx = make_uint4(0, 0, 0, 0);
        if (q < q1) {
            int ow = (int)(q % OW);
            long t2 = q / OW;
            int oh = (int)(t2 % OH);
            long n = t2 / OH;
            if (yv) ry = *reinterpret_cast<const uint4*>(dy + q * ldy + co0 + seg * 8);
            int ih = oh * stride + kh - pad, iw = ow * stride + kw - pad;
            if (xv && ih >= 0 && ih < H && iw >= 0 && iw < W)
                rx = *reinterpret_cast<const uint4*>(x + ((n * H + ih) * (long)W + iw) * ldx + ci0 + seg * 8);
        }
    };
    auto lstore = [&]() {
        const T* py = reinterpret_cast<const T*>(&ry);
        const T* px = reinterpret_cast<const T*>(&rx);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            lds_y[seg * 8 + j][pix] = py[j];
            lds_x[seg * 8 + j][pix] = px[j];
        }
    };

    const int wc = wave >> 1, wi = wave & 1;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (steps > 0) gload(0);
    for (int s = 0; s < steps; ++s) {
        __syncthreads();            // every wave is done reading the previous step's tiles
        lstore();
        __syncthreads();
        if (s + 1 < steps) gload(s + 1);
        frag fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const frag*>(&lds_y[wc * 32 + i * 16 + fr][fk]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const frag*>(&lds_x[wi * 32 + j * 16 + fr][fk]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = ops::mma(fa[i], fb[j], acc[i][j]);
    }

    // D rows = co ((lane>>4)*4 + r), cols = ci (lane & 15)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int ci = ci0 + wi * 32 + j * 16 + fr;
            if (ci >= Cin) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int co = co0 + wc * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (co < Cout) atomicAdd(dwp + (long)co * Kpad + tapi * Cin + ci, acc[i][j][r]);
            }
        }
}

template <typename T, int WGM, int WGN, int WM, int WN>
void launch_tile(const GeomDev& d, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                 hipStream_t st) {
    constexpr int BN = WGN * WN * 16;
    long pix = (long)d.N * d.Hg * d.Wg;
    int tm = (int)((pix + BM - 1) / BM), tn = (d.Cd + BN - 1) / BN;
    dim3 grid(tm * tn);
    int mode = d.Cs < BK ? 1 : (d.Cs % BK == 0 ? 2 : 0);
    if (mode == 2 && d.dma) mode = 3;
#define CONV_LAUNCH(ACC_, SM_)                                                                                        \
    hipLaunchKernelGGL((k_conv_mfma<T, WGM, WGN, WM, WN, ACC_, SM_>), grid, dim3(256), 0, st, d, (const T*)src,       \
                       (const T*)wm, bias, (T*)dst, tn)
    if (accumulate) { if (mode == 1) CONV_LAUNCH(true, 1); else if (mode == 2) CONV_LAUNCH(true, 2); else if (mode == 3) CONV_LAUNCH(true, 3); else CONV_LAUNCH(true, 0); }
    else { if (mode == 1) CONV_LAUNCH(false, 1); else if (mode == 2) CONV_LAUNCH(false, 2); else if (mode == 3) CONV_LAUNCH(false, 3); else CONV_LAUNCH(false, 0); }
#undef CONV_LAUNCH
}

// Tile width (tools/conv_tune.py on MI355X): the widest channel tile that still yields one workgroup per CU --
// small maps with many channels (20x20, K in the thousands) otherwise run ~100 workgroups through a 144-step
// K loop on a 256-CU chip; narrower tiles than that only add LDS reads per MFMA.  One-tap convs whose source
// stays in the 256 MB Infinity Cache prefer 64-wide tiles (re-reading the source per channel tile is cheap there).
int conv_tile_bn(const GeomDev& d_in) {
    const long tm = ((long)d_in.N * d_in.Hg * d_in.Wg + BM - 1) / BM;
    auto blocks = [&](int bn) { return tm * ((d_in.Cd + bn - 1) / bn); };
    int bn = d_in.Cd > 64 ? 128 : (d_in.Cd > 32 ? 64 : 32);
    const long src_bytes = (long)d_in.N * d_in.Hs * d_in.Ws * d_in.lds * 2;
    if (bn == 128 && d_in.ntaps == 1 && src_bytes <= (128L << 20)) bn = 64;
    while (bn > 32 && blocks(bn) < 256) bn >>= 1;
    const ConvTune& tu = conv_tune();                        // overrides: tuning runs and variant-forcing tests only
    if (tu.bn == 32 || tu.bn == 64 || tu.bn == 128) bn = tu.bn;
    return bn;
}

template <typename T>
void launch_conv_t(const GeomDev& d_in, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                   hipStream_t st) {
    const int bn = conv_tile_bn(d_in);
    GeomDev d = d_in;
    const ConvTune& tu = conv_tune();
    if (tu.tap_inner >= 0) d.tap_inner = tu.tap_inner;
    if (tu.dma >= 0) d.dma = tu.dma;
    if (bn == 128) launch_tile<T, 2, 2, 4, 4>(d, src, wm, bias, dst, accumulate, st);        // 128 x 128
    else if (bn == 64) launch_tile<T, 2, 2, 4, 2>(d, src, wm, bias, dst, accumulate, st);    // 128 x 64
    else launch_tile<T, 4, 1, 2, 2>(d, src, wm, bias, dst, accumulate, st);                  // 128 x 32
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int yolo_conv_tune_set(int bn, int tap_inner, int halo, int dma, int ring, int bm, int nst, int bk) {
    ConvTune& t = conv_tune();
    t.bn = bn; t.tap_inner = tap_inner; t.halo = halo; t.dma = dma; t.ring = ring; t.bm = bm; t.nst = nst; t.bk = bk;
    return YOLO_OK;
}

static int& wide_state() {
    static int v = [] { const char* e = getenv("YOLO_CONV_WIDE"); const int m = e ? atoi(e) : 2; return m < 0 ? 0 : m > 2 ? 2 : m; }();
    return v;
}
int conv_wide_flag() { return wide_state(); }
// test / A-B override of the 16-byte epilogue stores (store_pixel_blocks): 0 off, 1 exchange through ds_bpermute, 2 exchange
// through v_permlane16_swap
extern "C" int yolo_conv_wide_set(int on) { wide_state() = on < 0 ? 0 : on > 2 ? 2 : on; return YOLO_OK; }

int mfma_conv_eligible(const ConvGeom& g, int dtype, const void* src, const void* wm, const void* dst) {
    if (dtype != YOLO_BF16 && dtype != YOLO_F16) return 0;
    if (g.Cs % 8 || g.lds % 8 || g.Cd % 8 || g.ldd % 4) return 0;
    // 32-bit byte offsets / buffer descriptors in the gather, 32-bit pixel indices
    if ((long)g.N * g.Hs * g.Ws * g.lds + (long)(g.Ws + 1) * g.lds >= (1L << 30)) return 0;
    if ((long)g.N * g.Hg * g.Wg + 256 >= (1L << 31) || (long)g.Cd * g.Kpad >= (1L << 30)) return 0;
    if (!al16(src) || !al16(wm) || (reinterpret_cast<uintptr_t>(dst) & 7)) return 0;
    for (int t = 0; t < g.ntaps; ++t)
        if (g.dh[t] < -1 || g.dh[t] > 1 || g.dw[t] < -1 || g.dw[t] > 1) return 0;
    return 1;
}

// 3x3 stride-1 layers take the halo kernel (conv_halo.hip) when the map is large enough for its 8x16 / 16x16 pixel
// tiles to fill the chip; variant choice from tools/conv_tune.py.  yolo_conv_tune_set's third field overrides it
// (0 = gather kernel, 1..4 = halo variant) for tuning runs and the variant-forcing parity tests.
static int halo_variant(const ConvGeom& g) {
    const int v = conv_tune().halo;
    if (!halo_conv_eligible(g)) return 0;
    if (v >= 0 && v <= 4) return v;                          // 5..7 steer the row-block kernel only
    // measured in the training step (preset s, 32 images): the halo kernel wins on maps of 80x80 and more
    // (16x16-pixel tiles: 63 vs 81 us for 128->64 @80x80) and on 40x40 with exactly two 64-channel wave columns;
    // on smaller maps / other widths its tiles are too few or half empty and the gather kernel is level or better
    if ((long)g.Hg * g.Wg >= 80 * 80) return g.Cd >= 128 ? 2 : 3;
    if (g.Hg >= 40 && g.Wg >= 40 && g.Cd == 128) return 1;
    return 0;
}

// Maps 20 or 40 pixels wide: the row-block kernel (conv_rows.hip).  Returns 0 (not taken) or its variant (1 / 3 = 80 pixels
// x 64 channels per workgroup with four / three weight stages, 2 = 80 x 128, 4 = 160 x 64).  yolo_conv_tune_set's third
// field: 5 = default choice wherever eligible, 6 / 7 / 8 / 12 = force variant 1 / 2 / 3 / 4, 14 = variant 6 (10 rows of a
// 16-pixel-wide block x 64 channels, maps of any width), 0..4 and 9 = never.
// Default (tools/rows_bench.py, graph-replayed, 32 images, against the gather ring): 40-wide maps take the 160 x 64 tile
// (256->256: 83 -> 63 us, 64->64: 15.0 -> 12.8, 256->64: 38 -> 25), 20-wide maps too once the 80 x 64 tiling would put two
// workgroups on every CU (256->256: 35 -> 24 us), otherwise 80 x 64 (128->128: 17.8 -> 11.7, 512->64: 42.6 -> 21.5).
static int rows_variant(const ConvGeom& g) {
    const int v = conv_tune().halo;
    const int el = rows_conv_eligible(g);
    if (!el) return 0;
    if ((v >= 0 && v < 5) || v == 9) return 0;               // 9: this kernel off, everything else automatic (A/B runs)
    // narrow layers (fewer than 64 destination channels): 20 x 16 pixels x 32 channels.  Ahead of the gather kernels with a full
    // 32-channel source chunk (64->32 @80x80 forward 28.4 -> 21.0 us, its 32->64 data gradient 26.1 -> 18.6; 32->16 @160x160
    // forward 41.9 -> 36.5), behind them with a 16-channel source (half-empty chunks): 16 = forced, tests
    if (el == 3) return (v == 16 || ((v < 0 || v == 5) && g.Cs % 32 == 0)) ? 7 : 0;
    if (v == 14) return 6;                                   // 16-pixel-wide blocks, any map width
    // wider maps: 10 x 16-pixel blocks x 64 channels beat the halo kernel where the layer has exactly one 64-channel tile
    // (64->64 @80x80 35 -> 31 us forward, 31.5 -> 26.8 data gradient; 128->64 forward 51.5 -> 41.5; 64->64 @160x160 104 -> 97)
    if (el != 1) return (v < 0 || v == 5) && g.Cd == 64 && g.Cs >= 64 ? 6 : 0;          // 15: full-row blocks only (A/B)
    if (v == 6) return 1;
    if (v == 7) return g.Cd > 64 ? 2 : 1;
    if (v == 8) return 3;
    if (v == 12) return 4;
    if (g.Wg == 40) return 4;
    const long wgs = (long)g.N * ((g.Hg + 3) / 4) * ((g.Cd + 63) / 64);
    return wgs >= 512 ? 4 : 3;
}

int mfma_conv_plan(const ConvGeom& g, int dtype) {
    static const long long dummy[2] = {0, 0};
    if (const int rv = rows_variant(g)) return 4000 + rv;
    if (const int hv = halo_variant(g)) return 2000 + hv;
    if (ring_conv_eligible(g, dtype, dummy, dummy, dummy)) return ring_conv_plan(&g, 1);
    return 1000 + conv_tile_bn(to_dev(g));
}

int mfma_conv_launch(const ConvGeom& g, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                     int dtype, hipStream_t st) {
    if ((long)g.N * g.Hg * g.Wg == 0) return YOLO_OK;
    if (const int rv = rows_variant(g)) return rows_conv_launch(g, rv, src, wm, bias, dst, accumulate, dtype, st);
    if (const int hv = halo_variant(g)) return halo_conv_launch(g, hv, src, wm, bias, dst, accumulate, dtype, st);
    if (ring_conv_eligible(g, dtype, src, wm, dst)) {
        const long off0 = 0;
        return ring_conv_launch(&g, 1, &off0, (long)g.Cd * g.Kpad, src, wm, bias, dst, accumulate, dtype, st);
    }
    GeomDev d = to_dev(g);
    if (dtype == YOLO_BF16) launch_conv_t<bf16_t>(d, src, wm, bias, dst, accumulate, st);
    else launch_conv_t<f16_t>(d, src, wm, bias, dst, accumulate, st);
    return YOLO_LAUNCH_CHECK();
}

int mfma_wgrad_eligible(int Cin, int Cout, int ldx, int ldy, int dtype, const void* x, const void* dy) {
    if (dtype != YOLO_BF16 && dtype != YOLO_F16) return 0;
    if (Cin % 8 || Cout % 8 || ldx % 8 || ldy % 8) return 0;
    return al16(x) && al16(dy);
}

int mfma_wgrad_launch(const void* x, int ldx, const void* dy, int ldy, float* dwp, int Kpad, int N, int H, int W,
                      int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, hipStream_t st) {
    long P = (long)N * OH * OW;
    if (P == 0) return YOLO_OK;
    int cot = (Cout + 63) / 64, cit = (Cin + 63) / 64, nt = k * k;
    long tiles = (long)cot * cit * nt;
    long max_slabs = 4096 / tiles;
    if (max_slabs < 1) max_slabs = 1;
    long slab = (P + max_slabs - 1) / max_slabs;
    if (slab < 1024) slab = 1024;
    slab = (slab + WG_BP - 1) / WG_BP * WG_BP;
    int nslab = (int)((P + slab - 1) / slab);
    dim3 grid(cot, cit * nt, nslab);
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((k_wgrad_mfma<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, ldy,
                           dwp, Kpad, N, H, W, Cin, OH, OW, Cout, k, stride, cit, slab);
    else
        hipLaunchKernelGGL((k_wgrad_mfma<f16_t>), grid, dim3(256), 0, st, (const f16_t*)x, ldx, (const f16_t*)dy, ldy,
                           dwp, Kpad, N, H, W, Cin, OH, OW, Cout, k, stride, cit, slab);
    return YOLO_LAUNCH_CHECK();
}
