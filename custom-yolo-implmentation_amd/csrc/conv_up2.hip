// Data gradient of a 3x3 stride-2 convolution with the dy patch staged ONCE for all four parity classes.
//
// dx pixel (2a+ph, 2b+pw) sums 1, 2, 2 or 4 taps of dy around (a, b) (conv_geom.h: dgrad_s2_taps_1d).  The ring kernel
// runs the four classes as four gather GEMMs over the dy grid -- nine gathers of dy through the L2 -> LDS fill path
// (32->64 @320x320: 158 us against 63 us of HBM time).  Here a workgroup owns a TH x 16 block of the dy grid and ALL FOUR
// classes of it (a 2TH x 32 block of dx): per 32-channel chunk the (TH+1) x 17 dy patch goes to LDS once (64-byte pixel
// rows, chunk-swizzled), the nine (class, tap) products read it at scalar pixel offsets, and each class keeps its own
// accumulators (wave tile 64 pixels x 32 channels x 4 classes = 128 accumulator registers, two waves per SIMD).  Weights:
// the four class matrices of yolo_conv_pack_weights(mode 1), three (class, tap) tiles per barrier step through an LDS-DMA
// ring retired by counted s_waitcnt (the idiom of conv_ring.hip); the next chunk's patch rides the same queue.  The four
// classes of a pixel pair leave the same wave back to back, so L2 sees whole dx lines.
// PMC (128->128 @160x160, profiles/r2_up2_pmc.md): MFMA busy 25 % of the kernel, 59 % of wave time in s_waitcnt -- on the
// DMA queue, not on LDS (SQ_WAIT_INST_LDS 4 %, bank conflicts 0): 335 MB through HBM / L2 in 107 us, the kernel is
// memory-side bound like its neighbours; ring depth 3 = depth 4, fragment-read scheduling +-1 %.
#include "conv_dev.h"
#include <type_traits>

namespace {

constexpr int UPW = 17;        // patch width (16 + 1)

struct Up2Geom {
    int N, Hs, Ws, Cs, lds, Hd, Wd, Cd, ldd;
    int wm_off[4], Kpad[4];    // class matrices inside the packed buffer (elements)
    int wm_elems;
    // what store_pixel_blocks reads (no bias / activation / second source in a data gradient of this kind)
    int wide, act, ldr, ld2;
    const void* res;
    const void* acc2;
};

// the nine (class, tap) products in step order; taps of a class in conv_taps' order: (dh, dw) = (t / nw, t % nw)
__device__ __forceinline__ constexpr int up2_cls(int p) { return p == 0 ? 0 : p < 3 ? 1 : p < 5 ? 2 : 3; }
__device__ __forceinline__ constexpr int up2_tap(int p) { return p == 0 ? 0 : p < 3 ? p - 1 : p < 5 ? p - 3 : p - 5; }
__device__ __forceinline__ constexpr int up2_dh(int p) { return p == 4 || p >= 7 ? 1 : 0; }
__device__ __forceinline__ constexpr int up2_dw(int p) { return p == 2 || p == 6 || p == 8 ? 1 : 0; }

__device__ __forceinline__ void up2_dma(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// The counted wait in front of a step's barrier ALSO waits for this wave's own LDS reads (lgkmcnt(0)): the barrier declares the
// stage read in the previous step free, and hipcc software-pipelines fragment reads across a raw s_barrier (the reads are issued
// before it, their s_waitcnt lgkmcnt comes after it).  A refill that has to fetch from memory arrives long after such a read has
// executed; the zero-size-descriptor pieces issued "past the end of K" fetch nothing and can land first -- the read then returns
// zeros: one (chunk, tap) product missing from a whole workgroup tile, sporadically (found in k_dgrad2_patch, round 3: DESIGN
// section 6; tools/dbg_up2.py).
template <int N> __device__ __forceinline__ void up2_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory"); }

template <typename T, int TH, int BN, int WM, int NST, bool ACC>
__global__ __launch_bounds__((TH / WM) * (BN / 32) * 64) __attribute__((amdgpu_waves_per_eu(8 / WM, 8 / WM))) void k_dgrad2_patch(
    Up2Geom g, const T* __restrict__ src, const T* __restrict__ wm, T* __restrict__ dst, int tiles_h, int tiles_w, int ntile_n) {
    constexpr int WGM = TH / WM, WGN = BN / 32, NW = WGM * WGN;
    constexpr int PH = TH + 1, PPX = PH * UPW;
    constexpr int HP = ((PPX + 15) / 16 + NW - 1) / NW;      // patch pieces (16 pixels x 64 bytes) per wave and chunk
    constexpr int DW = (3 * BN / 16 + NW - 1) / NW;          // weight pieces (16 rows x 64 bytes) per wave and step: 3 tiles x BN rows
    constexpr int HBUF = NW * HP * 1024, STAGE = NW * DW * 1024;      // bytes: one patch buffer, one weight stage
    constexpr int OOB = (int)0x80000000;
    static_assert(NST == 3 || NST == 4, "ring depth");
    using ops = mfma_ops<T>;
    using frag = typename ops::frag;
    extern __shared__ __attribute__((aligned(1024))) char up2_smem[];          // [2][HBUF] patches, [NST][STAGE] weights
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)up2_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / ntile_n, tile_n = tile - tile_m * ntile_n;
    const int per_img = tiles_h * tiles_w;
    const int n = tile_m / per_img, trem = tile_m - n * per_img;
    const int ty = trem / tiles_w, tx = trem - ty * tiles_w;
    const int y0 = ty * TH, x0 = tx * 16;
    const int cd0 = tile_n * BN;

    // ---- DMA sources: fixed per-lane byte offsets (swizzle applied to the SOURCE chunk, the LDS image is lane-linear),
    // one scalar offset per chunk, out of range = zeros.  Patch rows: 64 bytes, chunk ^ ((pixel >> 1) & 3) -- conflict-free for
    // ds_read_b128's lane groups at any pixel offset; weight rows: k_conv_mfma's swizzle.
    const int a_bytes = g.N * g.Hs * g.Ws * g.lds * 2, w_bytes = g.wm_elems * 2;
    int hvoff[HP], wvoff[3][DW];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int px = (wave * HP + i) * 16 + (lane >> 2), ck = (lane & 3) ^ ((px >> 1) & 3);
        const int hy = px / UPW, hx = px - hy * UPW;
        const bool ok = px < PPX && y0 + hy < g.Hs && x0 + hx < g.Ws;
        hvoff[i] = ok ? ((hy * g.Ws + hx) * g.lds + ck * 8) * 2 : OOB;
    }
    const int hsoff0 = ((n * g.Hs + y0) * g.Ws + x0) * g.lds * 2;
#pragma unroll
    for (int j = 0; j < DW; ++j) {
        const int rs = (wave * DW + j) * 16 + (lane >> 2);   // row of the stage: (tile of the step, channel)
        const int pl = rs / BN, row = rs - pl * BN;
        const int kseg = (lane & 3) ^ ((-(row >> 2)) & 3);
        const bool ok = pl < 3 && cd0 + row < g.Cd;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int p = 3 * s + pl;                        // pl is a run-time value: the (class, tap) of product 3s + pl
            const int c = p == 0 ? 0 : p < 3 ? 1 : p < 5 ? 2 : 3;
            const int t = p == 0 ? 0 : p < 3 ? p - 1 : p < 5 ? p - 3 : p - 5;
            const int off = c == 0 ? g.wm_off[0] : c == 1 ? g.wm_off[1] : c == 2 ? g.wm_off[2] : g.wm_off[3];   // no dynamic kernarg indexing
            const int kp = c == 0 ? g.Kpad[0] : c == 1 ? g.Kpad[1] : c == 2 ? g.Kpad[2] : g.Kpad[3];
            wvoff[s][j] = ok ? (off + (cd0 + row) * kp + t * g.Cs + kseg * 8) * 2 : OOB;
        }
    }
    const int nchunk = g.Cs / BK;
    auto issue_patch = [&](int chunk) {                      // chunks past the end: zero-size descriptor, same piece count
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src), 0, chunk < nchunk ? a_bytes : 0, 0x00020000);
        const unsigned base = lds0 + (chunk & 1) * HBUF + wave * (HP * 1024);
#pragma unroll
        for (int i = 0; i < HP; ++i) up2_dma(rsa, base + i * 1024, hvoff[i], hsoff0 + chunk * 64);
    };
    int wchunk = 0, wbuf = 0;                                // uniform: chunk and ring slot of the next weight issue
    auto issue_w = [&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wm), 0, wchunk < nchunk ? w_bytes : 0, 0x00020000);
        const unsigned base = lds0 + 2 * HBUF + wbuf * STAGE + wave * (DW * 1024);
#pragma unroll
        for (int j = 0; j < DW; ++j) up2_dma(rsb, base + j * 1024, wvoff[s][j], wchunk * 64);
        if (s == 2) ++wchunk;
        if (++wbuf == NST) wbuf = 0;
    };

    // ---- compute state: wave (wgm, wgn) owns dy rows wgm*WM .. +WM-1 (16 pixels each) x dx channels wgn*32 .. +31, all classes
    const int wgm = wave / WGN, wgn = wave - wgm * WGN;
    const int crow = wgn * 32;
    const int fr = lane & 15, fg = lane >> 4;
    const int fk = (fg ^ ((-(fr >> 2)) & 3)) * 16;           // byte offset of this lane's swizzled weight chunk
    int apx[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) apx[i] = (wgm * WM + i) * UPW + fr;
    f32x4 acc[4][WM][2];                                     // [class][pixel tile][channel tile]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[c][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: patch 0, then NST-1 weight steps (the issue order of the steady state: step it issues step it+NST-1)
    issue_patch(0);
    issue_w(std::integral_constant<int, 0>{});
    issue_w(std::integral_constant<int, 1>{});
    if (NST == 4) issue_w(std::integral_constant<int, 2>{});
    int rbuf = 0;
    auto step = [&](auto s_c, int chunk) {
        constexpr int s = decltype(s_c)::value;
        // newer than this step's weights: NST-2 later steps, and the next patch when it went out in one of those iterations
        constexpr bool patch_newer = s != 0 && (NST == 4 || s == 1);
        up2_wait<(NST - 2) * DW + (patch_newer ? HP : 0)>();
        __builtin_amdgcn_s_barrier();                        // everyone's pieces of this step are in; last step's stage / patch is free
        if (s == 0) issue_patch(chunk + 1);
        issue_w(std::integral_constant<int, (s + NST - 1) % 3>{});
        const char* hp = up2_smem + (chunk & 1) * HBUF;
        const char* wp = up2_smem + 2 * HBUF + rbuf * STAGE;
        frag fa[3][2], fb[3][WM];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int p = 3 * s + pl;
            const int toff = up2_dh(p) * UPW + up2_dw(p);
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[pl][j] = *reinterpret_cast<const frag*>(wp + (pl * BN + crow + j * 16 + fr) * 64 + fk);
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int px = apx[i] + toff;
                fb[pl][i] = *reinterpret_cast<const frag*>(hp + px * 64 + ((fg ^ ((px >> 1) & 3)) << 4));
            }
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int c = up2_cls(3 * s + pl);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[c][i][j] = ops::mma(fa[pl][j], fb[pl][i], acc[c][i][j]);
        }
        // issue order inside the step: the fragment reads of product pl+1 before the MFMAs of product pl
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + WM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + WM, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * WM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + WM, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * WM, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * WM, 0);
        if (++rbuf == NST) rbuf = 0;
    };
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        step(std::integral_constant<int, 0>{}, chunk);
        step(std::integral_constant<int, 1>{}, chunk);
        step(std::integral_constant<int, 2>{}, chunk);
    }
    up2_wait<0>();                                           // the zero-fill pieces issued past the end of K

    // ---- epilogue: lane holds channels c..c+3 of dy-grid pixel (row wgm*WM+i, col fr); class (ph, pw) -> dx (2a+ph, 2b+pw)
    const int cq = fg * 4;
    const int b = x0 + fr;
    const int Hg = g.Hd >> 1, Wg = g.Wd >> 1;
    const float zero_bias[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int a = y0 + wgm * WM + i;
        const bool live = a < Hg && b < Wg;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long pix = live ? ((long)n * g.Hd + 2 * a + (c >> 1)) * (long)g.Wd + 2 * b + (c & 1) : 0;
            store_pixel_blocks<T, 2, ACC>(g, acc[c][i], zero_bias, dst, pix, live, cd0 + crow, cq, lane);
        }
    }
}

template <typename T, int TH, int BN, int WM, int NST>
int launch_up2(const Up2Geom& d, const void* src, const void* wm, void* dst, int accumulate, hipStream_t st) {
    constexpr int NW = (TH / WM) * (BN / 32), NTHR = NW * 64;
    constexpr int HP = (((TH + 1) * UPW + 15) / 16 + NW - 1) / NW, DW = (3 * BN / 16 + NW - 1) / NW;
    constexpr size_t lds = (size_t)(2 * NW * HP + NST * NW * DW) * 1024;
    const int th = (d.Hs + TH - 1) / TH, tw = (d.Ws + 15) / 16, tn = (d.Cd + BN - 1) / BN;
    const dim3 grid((unsigned)(d.N * th * tw * tn));
    if (accumulate) {
        static unsigned long long done = 0;         // per instantiation: devices that have the attribute
        if (int e = yolo_allow_dyn_lds(reinterpret_cast<const void*>(k_dgrad2_patch<T, TH, BN, WM, NST, true>), lds, done)) return e;
        hipLaunchKernelGGL((k_dgrad2_patch<T, TH, BN, WM, NST, true>), grid, dim3(NTHR), lds, st, d, (const T*)src, (const T*)wm, (T*)dst, th, tw, tn);
    } else {
        static unsigned long long done = 0;         // per instantiation: devices that have the attribute
        if (int e = yolo_allow_dyn_lds(reinterpret_cast<const void*>(k_dgrad2_patch<T, TH, BN, WM, NST, false>), lds, done)) return e;
        hipLaunchKernelGGL((k_dgrad2_patch<T, TH, BN, WM, NST, false>), grid, dim3(NTHR), lds, st, d, (const T*)src, (const T*)wm, (T*)dst, th, tw, tn);
    }
    return YOLO_LAUNCH_CHECK();
}

int up2_mode() {                                              // YOLO_DGRAD2_PATCH=0: never (A/B runs against the gather ring)
    static const int m = [] {
        const char* e = getenv("YOLO_DGRAD2_PATCH");
        return e ? atoi(e) : 1;
    }();
    return m;
}

}  // namespace

// gs[0..4): the four parity classes of one stride-2 data gradient (conv_generic.hip: dgrad_geom).  Returns the variant
// (0 = not taken, 8 = 8x16 dy pixels x 64 dx channels per workgroup, 16 = 16x16 x 32).
int up2_conv_variant(const ConvGeom* gs, int dtype) {
    static const int want_taps[4] = {1, 2, 2, 4};
    const ConvGeom& g = gs[0];
    const int mode = up2_mode();
    if (mode == 0 || conv_tune().ring > 0 || (conv_tune().halo >= 0 && conv_tune().halo <= 4)) return 0;   // forced-variant tests of the other kernels
    if (dtype != YOLO_BF16 && dtype != YOLO_F16) return 0;
    if ((g.Hd & 1) || (g.Wd & 1) || g.Hs * 2 != g.Hd || g.Ws * 2 != g.Wd || g.Cs % 32 || g.Cd % 8 || g.Cd < 32 || g.lds % 8 || g.ldd % 4) return 0;
    if ((long)g.N * g.Hs * g.Ws * g.lds >= (1L << 30) || (long)g.N * g.Hd * g.Wd >= (1L << 31)) return 0;
    for (int c = 0; c < 4; ++c) {
        const ConvGeom& q = gs[c];
        if (q.ntaps != want_taps[c] || q.ostep != 2 || q.sstride != 1 || q.ooff_h != (c >> 1) || q.ooff_w != (c & 1) || q.Hg != g.Hs ||
            q.Wg != g.Ws || q.acc2 != nullptr)
            return 0;
        const int nw = (c & 1) ? 2 : 1;
        for (int t = 0; t < q.ntaps; ++t)
            if (q.dh[t] != t / nw || q.dw[t] != t % nw) return 0;
    }
    // measured on every stride-2 layer of the step (tools/up2_bench.py, 32 images, ring -> here): 32->64 @320x320 157 -> 98 us,
    // 128->128 @160 145 -> 107, 256->256 @80 114 -> 88, 128->128 @80 50 -> 35, 256->256 @40 40 -> 35, 256->512 @40 65 -> 56;
    // with 16-byte stores (round 3): 91 / 99 / 86 / 32 / 36 / 56 us
    return g.Cd <= 32 ? 16 : 8;
}

int up2_conv_launch(const ConvGeom* gs, int variant, const long* wm_off, long wm_elems, const void* src, const void* wm, void* dst,
                    int accumulate, int dtype, hipStream_t st) {
    const ConvGeom& g = gs[0];
    if (wm_elems >= (1L << 30) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(wm) & 15) ||
        (reinterpret_cast<uintptr_t>(dst) & 7))
        return YOLO_ERR_ARG;
    Up2Geom d;
    d.N = g.N; d.Hs = g.Hs; d.Ws = g.Ws; d.Cs = g.Cs; d.lds = g.lds; d.Hd = g.Hd; d.Wd = g.Wd; d.Cd = g.Cd; d.ldd = g.ldd;
    for (int c = 0; c < 4; ++c) { d.wm_off[c] = (int)wm_off[c]; d.Kpad[c] = gs[c].Kpad; }
    d.wm_elems = (int)wm_elems;
    // (Round 3: with the 16-byte exchange path this kernel returned a whole workgroup tile of parity class (1,1) short of one
    // (chunk, tap) product in a few launches of a hundred.  Not the stores: the counted waits of the ring, see up2_wait.)
    d.wide = conv_wide_flag(); d.act = 0; d.res = nullptr; d.ldr = 0; d.acc2 = nullptr; d.ld2 = 0;
    if (g.N * g.Hs * g.Ws == 0) return YOLO_OK;
#define UP2_T(T_)                                                                                       \
    return variant == 16 ? launch_up2<T_, 16, 32, 4, 3>(d, src, wm, dst, accumulate, st)                \
                         : launch_up2<T_, 8, 64, 4, 3>(d, src, wm, dst, accumulate, st);
    if (dtype == YOLO_BF16) { UP2_T(bf16_t) }
    UP2_T(f16_t)
#undef UP2_T
}
