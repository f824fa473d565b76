// 3x3 (<= 9 taps, offsets in [-1,1]) stride-1 convolution on the matrix cores with the source patch staged ONCE.
//
// k_conv_mfma gathers its activation tile from global memory again for every tap, so the L2 -> CU path carries 9x
// the input.  Here a workgroup owns a TH x 16 block of output pixels of one image: per 32-channel chunk it stages the
// (TH+2) x 18 halo patch in LDS (96-byte pixel rows: 64 B payload + 32 B pad, which makes every ds_read_b128
// fragment read conflict-free at ANY pixel offset), and the taps are nine scalar offsets into that patch -- no
// address arithmetic, no second global read.  Weights stream through a double-buffered [BN][32] tile per (chunk, tap)
// exactly as in k_conv_mfma (64-byte rows, XOR chunk swizzle).  Forward and stride-1 data gradient share it
// (ConvGeom tap list + packed weights).  Wave tile 64 pixels (4 output rows) x 64 channels; waves = TH/4 x BN/64.
#include "conv_dev.h"

namespace {

constexpr int HROW = 48;       // elements per halo pixel row in LDS (32 payload + 16 pad) = 96 bytes
constexpr int HW = 18;         // halo width (16 + 2)

template <typename T, int TH, int BN, bool ACC>
__global__ __launch_bounds__((TH / 4) * (BN / 64) * 64) void k_conv_halo(GeomDev g, const T* __restrict__ src,
                                                                        const T* __restrict__ wm,
                                                                        const float* __restrict__ bias,
                                                                        T* __restrict__ dst, int tiles_h, int tiles_w,
                                                                        int ntile_n) {
    constexpr int WGM = TH / 4, WGN = BN / 64, NTHR = WGM * WGN * 64;
    constexpr int HH = TH + 2, HPX = HH * HW;
    constexpr int HR = (HPX * 4 + NTHR - 1) / NTHR;          // 16-byte halo chunks per thread
    constexpr int WR = (BN * 4) / NTHR;                      // 16-byte weight chunks per thread
    static_assert((BN * 4) % NTHR == 0 && WR >= 1, "weight tile must divide over the workgroup");
    using ops = mfma_ops<T>;
    using frag = typename ops::frag;
    __shared__ __attribute__((aligned(16))) T halo[2][HPX * HROW];
    __shared__ __attribute__((aligned(16))) T wl[2][BN][LDSROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / ntile_n, tile_n = tile - tile_m * ntile_n;
    const int per_img = tiles_h * tiles_w;
    const int n = tile_m / per_img, trem = tile_m - n * per_img;
    const int ty = trem / tiles_w, tx = trem - ty * tiles_w;
    const int y0 = ty * TH, x0 = tx * 16;
    const int cd0 = tile_n * BN;

    // ---- loaders (buffer descriptors: fixed per-lane byte offset, scalar per-step offset, out-of-range = 0)
    const int shift = (g.Ws + 1) * g.lds;
    const __amdgpu_buffer_rsrc_t rsa =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src) - shift, 0, (g.N * g.Hs * g.Ws * g.lds + shift) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wm), 0, g.Cd * g.Kpad * 2, 0x00020000);
    int hvoff[HR], hlds[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        const int h = tid + NTHR * r;
        const int px = h >> 2, ck = h & 3;
        const int hy = px / HW, hx = px - hy * HW;
        const bool ok = h < HPX * 4 && (unsigned)(y0 - 1 + hy) < (unsigned)g.Hs && (unsigned)(x0 - 1 + hx) < (unsigned)g.Ws;
        hvoff[r] = ok ? ((hy * g.Ws + hx) * g.lds + ck * 8) * 2 : (int)0x80000000;
        hlds[r] = h < HPX * 4 ? px * HROW + ck * 8 : -1;
    }
    // scalar origin of the patch: pixel (n, y0-1, x0-1) relative to the shifted descriptor base (never negative)
    const int hsoff0 = (((n * g.Hs + y0 - 1) * g.Ws + x0 - 1) * g.lds + shift) * 2;
    const int kseg = tid & 3, wrow0 = tid >> 2;
    const int sk = (kseg ^ ((-(wrow0 >> 2)) & 3)) * 8;       // swizzled chunk this thread stores (rows differ by NTHR/4: multiple of 16)
    int wvoff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = wrow0 + i * (NTHR / 4);
        wvoff[i] = (cd0 + row < g.Cd) ? ((cd0 + row) * g.Kpad + kseg * 8) * 2 : (int)0x80000000;
    }

    uint4 rh[HR], rw[WR];
    auto load_halo = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsa, hvoff[r], hsoff0 + chunk * 64, 0);
            rh[r] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto store_halo = [&](int buf) {
#pragma unroll
        for (int r = 0; r < HR; ++r)
            if (hlds[r] >= 0) *reinterpret_cast<uint4*>(&halo[buf][hlds[r]]) = rh[r];
    };
    auto load_w = [&](int wcol) {
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsb, wvoff[i], wcol * 2, 0);
            rw[i] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<uint4*>(&wl[buf][wrow0 + i * (NTHR / 4)][sk]) = rw[i];
    };

    // ---- compute state: wave (wgm, wgn) owns output rows wgm*4 .. +3 (16 pixels each) x channels wgn*64 .. +63
    const int wgm = wave / WGN, wgn = wave - wgm * WGN;
    const int crow = wgn * 64;
    const int fr = lane & 15, fg = lane >> 4;
    const int fk = (fg ^ ((-(fr >> 2)) & 3)) * 8;            // swizzled chunk of the weight fragment rows
    int abase[4];                                            // halo element offset of (output row i, col fr), tap (0,0)
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = ((wgm * 4 + i + 1) * HW + fr + 1) * HROW + fg * 8;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunk = g.Cs / BK;
    load_halo(0);
    load_w(0);
    store_halo(0);
    store_w(0);
    __syncthreads();
    int tap = 0, chunk = 0, it = 0;
    const int nit = nchunk * g.ntaps;
    for (; it < nit; ++it) {
        const int wb = it & 1, hb = chunk & 1;
        // next step's (chunk, tap)
        int ntap = tap + 1, nchk = chunk;
        if (ntap == g.ntaps) { ntap = 0; ++nchk; }
        const bool more = it + 1 < nit;
        const bool new_halo = more && ntap == 0;             // the next step starts a new channel chunk
        if (more) load_w(ntap * g.Cs + nchk * BK);
        if (new_halo) load_halo(nchk);
        const int toff = (((int)((g.dh_pack >> (2 * tap)) & 3u) - 1) * HW + ((int)((g.dw_pack >> (2 * tap)) & 3u) - 1)) * HROW;
        frag fa[4], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const frag*>(&wl[wb][crow + j * 16 + fr][fk]);
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const frag*>(&halo[hb][abase[i] + toff]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = ops::mma(fa[j], fb[i], acc[i][j]);
        if (more) store_w(wb ^ 1);
        if (new_halo) store_halo(hb ^ 1);
        __syncthreads();
        tap = ntap;
        chunk = nchk;
    }

    // ---- epilogue: lane holds channels c..c+3 of pixel (row wgm*4+i, col fr)
    const int cq = fg * 4;
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cd0 + crow + j * 16 + cq;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = (bias != nullptr && c < g.Cd) ? bias[c + r] : 0.f;
    }
    const int ox = x0 + fr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = y0 + wgm * 4 + i;
        const bool pv = oy < g.Hg && ox < g.Wg;
        store_pixel_blocks<T, 4, ACC>(g, acc[i], bv, dst, pv ? ((long)n * g.Hd + oy) * (long)g.Wd + ox : 0, pv, cd0 + crow, cq, lane);
        if (!pv) {
            // pixels outside the map must not reach the statistics
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- optional BatchNorm batch statistics of the stored (rounded) values, as in k_conv_mfma
    float* const stats = g.stats;
    if (stats != nullptr) {
        float* sacc = reinterpret_cast<float*>(&wl[0][0][0]);             // [2][BN]; LDS is idle after the K loop
        for (int t = tid; t < 2 * BN; t += NTHR) sacc[t] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = to_f<T>(from_f<T>(acc[i][j][r]));
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
        for (int t = tid; t < BN; t += NTHR)
            if (cd0 + t < g.Cd) {
                atomicAdd(o + cd0 + t, sacc[t]);
                atomicAdd(o + g.Cd + cd0 + t, sacc[BN + t]);
            }
    }
}

template <typename T, int TH, int BN>
void launch_halo(const GeomDev& d, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                 hipStream_t st) {
    constexpr int NTHR = (TH / 4) * (BN / 64) * 64;
    const int th = (d.Hg + TH - 1) / TH, tw = (d.Wg + 15) / 16, tn = (d.Cd + BN - 1) / BN;
    const dim3 grid((unsigned)(d.N * th * tw * tn));
    if (accumulate)
        hipLaunchKernelGGL((k_conv_halo<T, TH, BN, true>), grid, dim3(NTHR), 0, st, d, (const T*)src, (const T*)wm, bias,
                           (T*)dst, th, tw, tn);
    else
        hipLaunchKernelGGL((k_conv_halo<T, TH, BN, false>), grid, dim3(NTHR), 0, st, d, (const T*)src, (const T*)wm, bias,
                           (T*)dst, th, tw, tn);
}

}  // namespace

// Shapes the halo kernel takes: stride 1 in source and destination, 2..9 taps, channels a multiple of 32 (source)
// and at least 64 (destination).  variant: 1 = 8x16 pixels x 128 ch, 2 = 16x16 x 128, 3 = 16x16 x 64, 4 = 8x16 x 64.
int halo_conv_eligible(const ConvGeom& g) {
    return g.sstride == 1 && g.ostep == 1 && g.ooff_h == 0 && g.ooff_w == 0 && g.ntaps > 1 && g.Cs % 32 == 0 && g.Cd >= 64 &&
           g.Hg == g.Hs && g.Wg == g.Ws && g.Hd == g.Hg && g.Wd == g.Wg;
}

int halo_conv_launch(const ConvGeom& g, int variant, const void* src, const void* wm, const float* bias, void* dst,
                     int accumulate, int dtype, hipStream_t st) {
    const GeomDev d = to_dev(g);
#define HALO_T(T_)                                                                              \
    switch (variant) {                                                                          \
        case 1: launch_halo<T_, 8, 128>(d, src, wm, bias, dst, accumulate, st); break;          \
        case 2: launch_halo<T_, 16, 128>(d, src, wm, bias, dst, accumulate, st); break;         \
        case 3: launch_halo<T_, 16, 64>(d, src, wm, bias, dst, accumulate, st); break;          \
        default: launch_halo<T_, 8, 64>(d, src, wm, bias, dst, accumulate, st); break;          \
    }
    if (dtype == YOLO_BF16) { HALO_T(bf16_t) } else { HALO_T(f16_t) }
#undef HALO_T
    return YOLO_LAUNCH_CHECK();
}
