// 3x3 stride-1 convolution (forward and data gradient) for NARROW maps: 20 or 40 pixels wide (the stride-32 / stride-16
// stages at 640x640).  Those layers are a few hundred workgroups of one or two per CU, every operand byte is a first
// touch (L2 is cold at kernel start and all workgroups walk K in lockstep, so each weight slice misses for everyone at
// once) and a dependent round trip costs ~1-2 us: time = round trips in the K loop x that latency, divided by the bytes
// kept in flight.  The gather kernels (conv_ring.hip) keep the ring full but fetch every activation nine times; the
// 16-pixel-wide tiles of conv_halo.hip leave 17-37 % of a map row empty and prefetch one step.  Here:
//   * a workgroup owns 80 output pixels of one image as FULL ROWS (4 x 20 or 2 x 40): five 16-pixel MFMA tiles in linear
//     pixel order -- a tile may wrap from one row into the next, a lane's pixel only fixes its offset into the staged
//     patch and the taps stay nine scalar offsets;
//   * per 32-channel chunk the (rows + 2) x (W + 2) halo patch is staged ONCE, by LDS-DMA, two chunks deep (64-byte
//     pixel rows, the four 16-byte chunks XOR-swizzled by (pixel >> 1) & 3, patch rows W + 8 pixels apart: conflict-free for ds_read_b128's
//     lane groups at any pixel offset, also where a tile wraps into the next row);
//   * the four waves split the OUTPUT CHANNELS (16 or 32 each) and share the five pixel tiles, so the weight tile of a
//     step -- one kernel row = three taps x BN x 32 -- is the only per-step traffic; it runs through a ring of NST stages
//     filled by LDS-DMA and retired by counted s_waitcnt (the idiom of conv_ring.hip): NST-1 steps of weights plus the
//     next chunk's patch stay in flight across the step barriers.
// BatchNorm statistics, bias and the accumulate sources in the epilogue as in the other conv kernels.
#include "conv_dev.h"
#include <type_traits>

namespace {

__device__ __forceinline__ void rows_dma(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// The counted wait in front of a step's barrier ALSO waits for this wave's own LDS reads (lgkmcnt(0)): the barrier declares the
// stage read in the previous step free, and hipcc software-pipelines fragment reads across a raw s_barrier (the reads are issued
// before it, their s_waitcnt lgkmcnt comes after it).  A refill that has to fetch from memory arrives long after such a read has
// executed; the zero-size-descriptor pieces issued "past the end of K" fetch nothing and can land first -- the read then returns
// zeros: one (chunk, tap) product missing from a whole workgroup tile, sporadically (found in k_dgrad2_patch, round 3: DESIGN
// section 6; tools/dbg_up2.py).
template <int N> __device__ __forceinline__ void rows_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory"); }

template <typename T, int W, int WGM, int NWAVE, int WN, int NST, bool ACC>
__global__ __launch_bounds__(64 * NWAVE) void k_conv_rows(GeomDev g, const T* __restrict__ src, const T* __restrict__ wm,
                                                   const float* __restrict__ bias, T* __restrict__ dst, int tiles_h,
                                                   int tiles_w, int ntile_n) {
    // W == 16: blocks of 16-pixel rows at column x0 of a map of any width (a pixel tile is one block row, never wraps: pitch 18);
    // otherwise the block rows ARE map rows (tiles_w == 1, x0 == 0)
    constexpr int R = WGM * 80 / W, HWID = W == 16 ? 18 : W + 8, HH = R + 2, HPX = HH * HWID;     // row pitch W + 8: a tile that wraps into the next row keeps the bank pattern
    constexpr int WGN = NWAVE / WGM;                         // waves: WGM groups of five pixel tiles x WGN channel groups
    constexpr int NTHR = 64 * NWAVE, BN = WGN * WN * 16;
    constexpr int HP = ((HPX + 15) / 16 + NWAVE - 1) / NWAVE; // patch pieces (16 pixels x 64 bytes) per wave
    constexpr int DW = (3 * BN / 16 + NWAVE - 1) / NWAVE;    // weight pieces (16 rows x 64 bytes) per wave and step: 3 taps x BN rows (+ idle pieces)
    constexpr int HBUF = NWAVE * HP * 1024, STAGE = NWAVE * DW * 1024;     // bytes: one patch buffer, one weight stage
    constexpr int OOB = (int)0x80000000;
    static_assert(R * W == WGM * 80 && NST >= 3 && NST <= 4 && NWAVE == 4, "block shape");
    using ops = mfma_ops<T>;
    using frag = typename ops::frag;
    extern __shared__ __attribute__((aligned(1024))) char rows_smem[];        // [2][HBUF] patches, [NST][STAGE] weights
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)rows_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / ntile_n, tile_n = tile - tile_m * ntile_n;
    const int per_img = tiles_h * tiles_w;
    const int n = tile_m / per_img, trem = tile_m - n * per_img;
    const int ty = trem / tiles_w, tx = trem - ty * tiles_w;
    const int y0 = ty * R, x0 = tx * W;
    const int cd0 = tile_n * BN;

    // ---- DMA sources: fixed per-lane byte offsets (swizzle applied to the SOURCE chunk, the LDS image is lane-linear),
    // one scalar offset per chunk / step, out of range = zeros
    const int shift = (g.Ws + 1) * g.lds;
    const int a_bytes = (g.N * g.Hs * g.Ws * g.lds + shift) * 2, w_bytes = g.Cd * g.Kpad * 2;
    int hvoff[HP], wvoff[DW];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int px = (wave * HP + i) * 16 + (lane >> 2), ck = (lane & 3) ^ ((px >> 1) & 3);
        const int hy = px / HWID, hx = px - hy * HWID;
        const bool ok = px < HPX && (unsigned)(y0 - 1 + hy) < (unsigned)g.Hs && (unsigned)(x0 + hx - 1) < (unsigned)g.Ws;
        hvoff[i] = (ok && ck * 8 < g.Cs) ? ((hy * g.Ws + hx) * g.lds + ck * 8) * 2 : OOB;      // ck*8 >= Cs: a 16-channel source fills half a chunk
    }
    // scalar origin of the patch: pixel (n, y0-1, -1) relative to the shifted descriptor base (never negative)
    const int hsoff0 = (((n * g.Hs + y0 - 1) * g.Ws + x0 - 1) * g.lds + shift) * 2;
#pragma unroll
    for (int j = 0; j < DW; ++j) {
        const int rs = (wave * DW + j) * 16 + (lane >> 2);   // row of the stage: (tap of the step, channel)
        const int tl = rs / BN, row = rs - tl * BN;
        const int kseg = (lane & 3) ^ ((-(row >> 2)) & 3);   // k_conv_mfma's swizzle
        wvoff[j] = (tl < 3 && cd0 + row < g.Cd && kseg * 8 < g.Cs) ? ((cd0 + row) * g.Kpad + tl * g.Cs + kseg * 8) * 2 : OOB;
    }
    const int nchunk = (g.Cs + BK - 1) / BK, nit = nchunk * 3;   // Cs is a multiple of 32, or 16 (one half-filled chunk)
    auto issue_halo = [&](int chunk) {                       // chunks past the end: zero-size descriptor, same piece count
        const __amdgpu_buffer_rsrc_t rsa =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src) - shift, 0, chunk < nchunk ? a_bytes : 0, 0x00020000);
        const unsigned base = lds0 + (chunk & 1) * HBUF + wave * (HP * 1024);
#pragma unroll
        for (int i = 0; i < HP; ++i) rows_dma(rsa, base + i * 1024, hvoff[i], hsoff0 + chunk * 64);
    };
    int wstep = 0, wchunk = 0, wrow = 0, wbuf = 0;           // uniform: the step the next issue_w() fetches
    auto issue_w = [&]() {
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wm), 0, wstep < nit ? w_bytes : 0, 0x00020000);
        const unsigned base = lds0 + 2 * HBUF + wbuf * STAGE + wave * (DW * 1024);
        const int soff = (3 * wrow * g.Cs + wchunk * BK) * 2;
#pragma unroll
        for (int j = 0; j < DW; ++j) rows_dma(rsb, base + j * 1024, wvoff[j], soff);
        ++wstep;
        if (++wrow == 3) { wrow = 0; ++wchunk; }
        if (++wbuf == NST) wbuf = 0;
    };

    // ---- compute state: wave (wgm, wgn) works on pixel tiles wgm*5 .. +4 and owns channels wgn*WN*16 .. +WN*16
    const int wgm = wave / WGN, wgn = wave - wgm * WGN;
    const int crow = wgn * WN * 16;
    const int fr = lane & 15, fg = lane >> 4;
    const int fk = (fg ^ ((-(fr >> 2)) & 3)) * 16;           // byte offset of this lane's swizzled weight chunk
    // swizzled byte offset of this lane's fragment of (pixel tile i, tap) inside a patch buffer: 45 registers instead of five
    // address instructions per read (PMC: 5.7 VALU per MFMA before, and one wave per SIMD issues them in line with its MFMAs)
    int aoff[5][9];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = (wgm * 5 + i) * 16 + fr, pr = p / W, pc = p - pr * W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int px = (pr + (int)((g.dh_pack >> (2 * tap)) & 3u)) * HWID + pc + (int)((g.dw_pack >> (2 * tap)) & 3u);
            aoff[i][tap] = px * 64 + ((fg ^ ((px >> 1) & 3)) << 4);
        }
    }
    const int woff = (crow + fr) * 64 + fk;                  // this lane's byte offset inside a 16-row weight block
    f32x4 acc[5][WN];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_halo(0);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue_w();
    int rbuf = 0;
    auto step = [&](auto srow_c, auto par_c, int chunk) {
        constexpr int srow = decltype(srow_c)::value, par = decltype(par_c)::value;   // par: patch buffer of this chunk (static: folds into the read's immediate)
        // newer than this step's weights: NST-2 later steps, and the next patch when it went out in one of those iterations
        constexpr bool patch_newer = srow != 0 && (NST == 4 || srow == 1);
        rows_wait<(NST - 2) * DW + (patch_newer ? HP : 0)>();
        __builtin_amdgcn_s_barrier();                        // everyone's pieces of this step are in; the stage / patch read last step is free
        if (srow == 0) issue_halo(chunk + 1);
        issue_w();
        const char* hp = rows_smem + par * HBUF;
        // three stages, three steps per chunk: the ring slot of a step is its kernel row -- static as well
        const char* wp = rows_smem + 2 * HBUF + (NST == 3 ? srow : rbuf) * STAGE;
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            const int tap = 3 * srow + tl;
            frag fa[WN], fb[5];
#pragma unroll
            for (int j = 0; j < WN; ++j) fa[j] = *reinterpret_cast<const frag*>(wp + (tl * BN + j * 16) * 64 + woff);
#pragma unroll
            for (int i = 0; i < 5; ++i) fb[i] = *reinterpret_cast<const frag*>(hp + aoff[i][tap]);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = ops::mma(fa[j], fb[i], acc[i][j]);
        }
        if (++rbuf == NST) rbuf = 0;
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    int chunk = 0;
    for (; chunk + 1 < nchunk; chunk += 2) {
        step(c0{}, c0{}, chunk); step(c1{}, c0{}, chunk); step(c2{}, c0{}, chunk);
        step(c0{}, c1{}, chunk + 1); step(c1{}, c1{}, chunk + 1); step(c2{}, c1{}, chunk + 1);
    }
    if (chunk < nchunk) { step(c0{}, c0{}, chunk); step(c1{}, c0{}, chunk); step(c2{}, c0{}, chunk); }
    rows_wait<0>();                                          // the zero-fill pieces issued past the end of K
    __syncthreads();
    T* const wl = reinterpret_cast<T*>(rows_smem);           // LDS is idle from here on (statistics scratch)

    // ---- epilogue: lane holds channels c..c+3 of its pixel of tile i
    const int cq = fg * 4;
    float bv[WN][4];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int c = cd0 + crow + j * 16 + cq;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = (bias != nullptr && c < g.Cd) ? bias[c + r] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = (wgm * 5 + i) * 16 + fr, pr = p / W, pc = p - pr * W;
        const int oy = y0 + pr, ox = x0 + pc;
        const bool live = oy < g.Hg && ox < g.Wg;
        store_pixel_blocks<T, WN, ACC>(g, acc[i], bv, dst, live ? ((long)n * g.Hd + oy) * (long)g.Wd + ox : 0, live, cd0 + crow, cq, lane);
        if (!live) {
            // pixels outside the map must not reach the statistics
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- optional BatchNorm batch statistics of the stored (rounded) values, as in k_conv_mfma
    float* const stats = g.stats;
    if (stats != nullptr) {
        float* sacc = reinterpret_cast<float*>(wl);          // [2][BN]; LDS is idle after the K loop (last barrier passed)
        for (int t = tid; t < 2 * BN; t += NTHR) sacc[t] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float s[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 5; ++i)
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

template <typename T, int W, int WGM, int NWAVE, int WN, int NST>
int launch_rows(const GeomDev& d, const void* src, const void* wm, const float* bias, void* dst, int accumulate, hipStream_t st) {
    constexpr int R = WGM * 80 / W, BN = (NWAVE / WGM) * WN * 16, HPX = (R + 2) * (W == 16 ? 18 : W + 8);
    constexpr size_t lds = 2 * (size_t)(NWAVE * (((HPX + 15) / 16 + NWAVE - 1) / NWAVE)) * 1024 + (size_t)NST * NWAVE * ((3 * BN / 16 + NWAVE - 1) / NWAVE) * 1024;
    const int th = (d.Hg + R - 1) / R, tw = W == 16 ? (d.Wg + 15) / 16 : 1, tn = (d.Cd + BN - 1) / BN;
    const dim3 grid((unsigned)(d.N * th * tw * tn));
    if (accumulate) {
        static unsigned long long done = 0;         // per instantiation: devices that have the attribute
        if (int e = yolo_allow_dyn_lds(reinterpret_cast<const void*>(k_conv_rows<T, W, WGM, NWAVE, WN, NST, true>), lds, done)) return e;
        hipLaunchKernelGGL((k_conv_rows<T, W, WGM, NWAVE, WN, NST, true>), grid, dim3(64 * NWAVE), lds, st, d, (const T*)src, (const T*)wm, bias, (T*)dst, th, tw, tn);
    } else {
        static unsigned long long done = 0;         // per instantiation: devices that have the attribute
        if (int e = yolo_allow_dyn_lds(reinterpret_cast<const void*>(k_conv_rows<T, W, WGM, NWAVE, WN, NST, false>), lds, done)) return e;
        hipLaunchKernelGGL((k_conv_rows<T, W, WGM, NWAVE, WN, NST, false>), grid, dim3(64 * NWAVE), lds, st, d, (const T*)src, (const T*)wm, bias, (T*)dst, th, tw, tn);
    }
    return YOLO_LAUNCH_CHECK();
}

}  // namespace

// Shapes the row-block kernel takes: the nine taps of a 3x3 conv, stride 1 in source and destination, maps exactly 20 or
// 40 pixels wide, source channels a multiple of 32, at least 64 destination channels.
// rows_conv_eligible: 1 = full-row blocks (20- / 40-wide maps), 2 = only the 16-pixel-wide blocks (variant 6), 3 = only the
// 16-pixel-wide blocks with a 32-channel tile (variant 7: fewer than 64 destination channels or a 16-channel source), 0 = none
int rows_conv_eligible(const ConvGeom& g) {
    if (!(g.sstride == 1 && g.ostep == 1 && g.ooff_h == 0 && g.ooff_w == 0 && g.ntaps == 9 && (g.Cs % 32 == 0 || g.Cs == 16) && g.Cd >= 16 &&
          g.Cd % 8 == 0 && g.Hg == g.Hs && g.Wg == g.Ws && g.Hd == g.Hg && g.Wd == g.Wg))
        return 0;
    for (int t = 0; t < 9; ++t)
        if (g.dh[t] < -1 || g.dh[t] > 1 || g.dw[t] < -1 || g.dw[t] > 1) return 0;
    if (g.Cd < 64 || g.Cs == 16) return 3;                   // narrow layers: variant 7 only
    return (g.Wg == 20 || g.Wg == 40) ? 1 : 2;
}

int rows_conv_launch(const ConvGeom& g, int variant, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                     int dtype, hipStream_t st) {
    const GeomDev d = to_dev(g);
    // variant: 80 pixels x 64 channels, 16 channels per wave: 1 = four weight stages, 3 = three; 2 = 80 x 128 (32 per wave);
    // 4 = 160 pixels x 64 channels: two pixel groups x two channel groups of waves, wave tile 5 x 2.
    // Measured with the compute or the DMA compiled out (256->256 @40x40, 80 x 64 tile, 87 us): DMA ring + barriers alone 64 us
    // (60 GB/s per CU through the L2 -> LDS fill path, three quarters of it weights), fragment reads + MFMAs + barriers alone
    // 63 us -- both sides bound it.  Twice the pixels per workgroup halves the weight fill per output AND the LDS reads per
    // MFMA (0.7 instead of 1.2): 62 us, and every 40-wide layer gains; 160 x 128 (one workgroup per CU) and two-wave
    // workgroups lose again.
#define ROWS_W(T_, W_)                                                                                               \
    switch (variant) {                                                                                               \
        case 2: return launch_rows<T_, W_, 1, 4, 2, 3>(d, src, wm, bias, dst, accumulate, st);                          \
        case 3: return launch_rows<T_, W_, 1, 4, 1, 3>(d, src, wm, bias, dst, accumulate, st);                       \
        case 4: return launch_rows<T_, W_, 2, 4, 2, 3>(d, src, wm, bias, dst, accumulate, st);                       \
        default: return launch_rows<T_, W_, 1, 4, 1, 4>(d, src, wm, bias, dst, accumulate, st);                         \
    }
#define ROWS_T(T_)                                                                                                   \
    if (variant == 6) return launch_rows<T_, 16, 2, 4, 2, 3>(d, src, wm, bias, dst, accumulate, st);   /* 10 x 16 pixels x 64 ch */ \
    if (variant == 7) return launch_rows<T_, 16, 4, 4, 2, 3>(d, src, wm, bias, dst, accumulate, st);   /* 20 x 16 pixels x 32 ch */ \
    if (g.Wg == 20) { ROWS_W(T_, 20) }                                                                               \
    ROWS_W(T_, 40)
    if (dtype == YOLO_BF16) { ROWS_T(bf16_t) }
    ROWS_T(f16_t)
#undef ROWS_T
#undef ROWS_W
}
