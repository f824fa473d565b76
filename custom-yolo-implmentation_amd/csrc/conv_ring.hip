// bf16 / f16 implicit-GEMM convolution (forward and data gradient) with a software-pipelined LDS ring.
//
// k_conv_mfma (conv_mfma.hip) issues the next K-step's tiles, then waits for ALL of them before it reads the current
// ones: hipcc cannot tell which LDS bytes a `buffer_load ... lds` builtin writes, so it puts `s_waitcnt vmcnt(0)` in
// front of the first ds_read of every step -- inside one workgroup nothing overlaps, and layers with one workgroup per
// CU or fewer (every 20x20 / 40x40 map of the model) run at one L2 round trip per 32-deep K-step.  Here:
//   * K-steps of 64 elements (two MFMA k-blocks per barrier, LDS rows of 128 bytes = one full cache line per pixel row
//     and step, the eight 16-byte chunks of a row XOR-swizzled by (row >> 1) & 7) or of 32 elements (64-byte rows,
//     k_conv_mfma's swizzle): conflict-free ds_read_b128 fragment reads; the swizzle is applied to the lane's SOURCE
//     chunk, the LDS-DMA image is lane-linear;
//   * a ring of NST stages filled by LDS-DMA issued from inline asm (invisible to hipcc's wait insertion) and retired
//     by a COUNTED `s_waitcnt vmcnt((NST-2) * loads per stage)` + one raw s_barrier per step: NST-1 steps of loads
//     stay in flight across barriers.  Stages past the end of K are issued with out-of-range offsets (the hardware
//     writes zeros, no memory traffic), so the count is the same in every iteration;
//   * both operands through buffer descriptors as in k_conv_mfma: fixed per-lane byte offset, one scalar offset per
//     step (tap, channel chunk), padding taps / rows past the end / partial channel chunks = offset out of range = 0;
//   * up to four sub-problems per launch (the four parity classes of a stride-2 data gradient: own tap list, weight
//     matrix and destination phase) -- one launch of ~4x the workgroups instead of four small ones.
// Source channels: a multiple of 32, at least 64 (narrower layers stay on k_conv_mfma).  MFMA roles are swapped
// (A = weights, B = activations): a lane's four accumulator registers are four consecutive output channels.
#include "conv_dev.h"

int halo_conv_eligible(const ConvGeom& g);

namespace {

struct RingSub {
    int Hg, Wg, ooff_h, ooff_w, ntaps, KT, Kpad, wm_off, tile0, npix;
    unsigned dh_pack, dw_pack;
};
struct RingGeom {
    int N, Hs, Ws, Cs, lds, Hd, Wd, Cd, ldd, ostep, sstride, nsub, spt, ntile_n, wm_elems;
    float* stats;
    const void* acc2;           // ACC launches: second accumulate source (row stride ld2) or null
    int ld2;
    int wide;                   // 16-byte epilogue stores (see store_pixel_blocks)
    int act;                    // inference epilogue (see ConvGeom)
    const void* res;
    int ldr;
    RingSub sub[4];
};

// one LDS-DMA piece: 64 lanes x 16 bytes from the descriptor (voffset per lane + scalar offset) to LDS at lds_addr + 16*lane
__device__ __forceinline__ void ring_dma(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// The counted wait in front of a step's barrier ALSO waits for this wave's own LDS reads (lgkmcnt(0)): the barrier declares the
// stage read in the previous step free, and hipcc software-pipelines fragment reads across a raw s_barrier (the reads are issued
// before it, their s_waitcnt lgkmcnt comes after it).  A refill that has to fetch from memory arrives long after such a read has
// executed; the zero-size-descriptor pieces issued "past the end of K" fetch nothing and can land first -- the read then returns
// zeros: one (chunk, tap) product missing from a whole workgroup tile, sporadically (found in k_dgrad2_patch, round 3: DESIGN
// section 6; tools/dbg_up2.py).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory"); }

// RB = bytes per LDS row = bytes of K per step: 128 (64-deep steps, 8 rows per piece, chunk swizzle (row >> 1) & 7) or
// 64 (32-deep steps, 16 rows per piece, the swizzle of k_conv_mfma: (-(row >> 2)) & 3)
template <int RB> __device__ __forceinline__ int ring_swz(int row) {
    if constexpr (RB == 128) return (row >> 1) & 7;
    else return (-(row >> 2)) & 3;
}

template <typename T, int RB, int WGM, int WGN, int WM, int WN, int NST, bool ACC>
__global__ __launch_bounds__(256) void k_conv_ring(RingGeom g, const T* __restrict__ src, const T* __restrict__ wm,
                                                   const float* __restrict__ bias, T* __restrict__ dst) {
    static_assert(WGM * WGN == 4 && (RB == 64 || RB == 128), "four waves; 32- or 64-deep steps");
    constexpr int BM_ = WGM * WM * 16, BN = WGN * WN * 16;
    constexpr int KE = RB / 2, SLOTS = RB / 16, RPP = 64 / SLOTS;     // K elements per step; 16-byte chunks per row; rows per piece
    constexpr int LA = BM_ / (4 * RPP), LB = BN / (4 * RPP);          // LDS-DMA pieces per wave and stage
    static_assert(LA >= 1 && LB >= 1 && BM_ % (4 * RPP) == 0 && BN % (4 * RPP) == 0, "tile shape");
    constexpr int L = LA + LB;
    constexpr int A_BYTES = BM_ * RB, STAGE = (BM_ + BN) * RB;
    constexpr int OOB = (int)0x80000000;
    using ops = mfma_ops<T>;
    using frag = typename ops::frag;
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / g.ntile_n, tile_n = tile - tile_m * g.ntile_n;
    int cls = 0;
    for (int i = 1; i < g.nsub; ++i)
        if (tile_m >= g.sub[i].tile0) cls = i;
    const RingSub sb = g.sub[cls];
    const int m0 = (tile_m - sb.tile0) * BM_;
    const int cd0 = tile_n * BN;

    // ---- loader state.  Wave w fills rows [w*BM/4, (w+1)*BM/4) of the activation tile and [w*BN/4, ...) of the weight
    // tile, RPP rows per piece: lane = (row in piece) * SLOTS + slot, and the chunk that belongs in slot s of row r is
    // s ^ swz(r).  A row past the last pixel / channel gets an out-of-range offset once and for all.
    const int rp = lane / SLOTS, slot = lane % SLOTS;
    int voffa[LA], voffb[LB];
    unsigned vmask[LA];                                       // bit t: tap t of this row is inside the source image
    unsigned tailok = 0;                                      // bit i / bit 8+j: the lane's chunk exists in a partial last chunk step
    const int spt = (g.Cs + KE - 1) / KE;                     // steps per tap
    const int tail_c0 = (spt - 1) * KE;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int row = wave * (BM_ / 4) + i * RPP + rp;
        const int chunk = slot ^ ring_swz<RB>(row);
        const int q = m0 + row;
        const bool rv = q < sb.npix;
        const unsigned qq = rv ? q : 0;
        const unsigned t2 = qq / (unsigned)sb.Wg, b = qq - t2 * sb.Wg;
        const unsigned n = t2 / (unsigned)sb.Hg, a = t2 - n * sb.Hg;
        const int hs0 = (int)a * g.sstride, ws0 = (int)b * g.sstride;
        voffa[i] = rv ? ((((int)n * g.Hs + hs0) * g.Ws + ws0) * g.lds + chunk * 8) * 2 : OOB;
        unsigned m = 0;
        for (int t = 0; t < sb.ntaps; ++t) {
            const int dh = (int)((sb.dh_pack >> (2 * t)) & 3u) - 1, dw = (int)((sb.dw_pack >> (2 * t)) & 3u) - 1;
            const bool ok = (unsigned)(hs0 + dh) < (unsigned)g.Hs && (unsigned)(ws0 + dw) < (unsigned)g.Ws;
            m |= (ok ? 1u : 0u) << t;
        }
        vmask[i] = m;
        tailok |= (tail_c0 + chunk * 8 < g.Cs ? 1u : 0u) << i;
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int row = wave * (BN / 4) + j * RPP + rp;
        const int chunk = slot ^ ring_swz<RB>(row);
        voffb[j] = (cd0 + row < g.Cd) ? (sb.wm_off + (cd0 + row) * sb.Kpad + chunk * 8) * 2 : OOB;
        tailok |= (tail_c0 + chunk * 8 < g.Cs ? 1u : 0u) << (8 + j);
    }
    // descriptor base = src - one row - one pixel, so the scalar tap offset (dh+1, dw+1) is never negative
    const int shift = (g.Ws + 1) * g.lds;
    const int a_bytes = (g.N * g.Hs * g.Ws * g.lds + shift) * 2, w_bytes = g.wm_elems * 2;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const bool partial = (g.Cs % KE) != 0;
    const bool taps = sb.ntaps > 1;                           // a one-tap conv has no padding taps: nothing per lane and step

    int tap = 0, cstep = 0, issued = 0, wbuf = 0;             // uniform: the step the next issue() fetches
    auto issue = [&]() {
        // steps past the end of K go out through descriptors of zero records: every lane is out of range, LDS gets
        // zeros, nothing is fetched -- and every iteration issues the same number of pieces
        const bool live = issued < sb.KT;
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src) - shift, 0, live ? a_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wm), 0, live ? w_bytes : 0, 0x00020000);
        const int t = live ? tap : 0;
        const int soffa = (((int)((sb.dh_pack >> (2 * t)) & 3u) * g.Ws + (int)((sb.dw_pack >> (2 * t)) & 3u)) * g.lds + cstep * KE) * 2;
        const int soffb = (t * g.Cs + cstep * KE) * 2;
        const unsigned base = lds0 + wbuf * STAGE + wave * (LA * 1024);
        const unsigned baseb = lds0 + wbuf * STAGE + A_BYTES + wave * (LB * 1024);
        if (partial && cstep == spt - 1) {                    // last chunk step of a tap, channel count not a multiple of the step
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                const bool ok = ((vmask[i] >> t) & 1u) && ((tailok >> i) & 1u);
                ring_dma(rsa, base + i * 1024, ok ? voffa[i] : OOB, soffa);
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) ring_dma(rsb, baseb + j * 1024, ((tailok >> (8 + j)) & 1u) ? voffb[j] : OOB, soffb);
        } else if (taps) {
            const unsigned bit = 1u << t;
#pragma unroll
            for (int i = 0; i < LA; ++i) ring_dma(rsa, base + i * 1024, (vmask[i] & bit) ? voffa[i] : OOB, soffa);
#pragma unroll
            for (int j = 0; j < LB; ++j) ring_dma(rsb, baseb + j * 1024, voffb[j], soffb);
        } else {
#pragma unroll
            for (int i = 0; i < LA; ++i) ring_dma(rsa, base + i * 1024, voffa[i], soffa);
#pragma unroll
            for (int j = 0; j < LB; ++j) ring_dma(rsb, baseb + j * 1024, voffb[j], soffb);
        }
        ++issued;
        if (++cstep == spt) { cstep = 0; ++tap; }
        if (++wbuf == NST) wbuf = 0;
    };

    // ---- compute state
    const int wgm = wave / WGN, wgn = wave - wgm * WGN;
    const int prow = wgm * WM * 16, crow = wgn * WN * 16;
    const int fr = lane & 15, fg = lane >> 4;
    const int fsw = ring_swz<RB>(fr);                         // swizzle of this lane's fragment rows (row bases are multiples of 16)
    const int foff0 = fr * RB + ((fg ^ fsw) * 16), foff1 = fr * RB + (((4 + fg) ^ fsw) * 16);
    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue();
    int rbuf = 0;
    for (int kt = 0; kt < sb.KT; ++kt) {
        wait_vmcnt<(NST - 2) * L>();                          // this wave's pieces of stage kt have landed ...
        __builtin_amdgcn_s_barrier();                         // ... and so have everyone else's; stage kt-1 is free again
        issue();                                              // refill the stage read in the previous step
        const char* sa = smem + rbuf * STAGE + prow * RB;
        const char* sw = smem + rbuf * STAGE + A_BYTES + crow * RB;
#pragma unroll
        for (int ks = 0; ks < RB / 64; ++ks) {
            const int fo = ks ? foff1 : foff0;
            frag fa[WN], fb[WM];
#pragma unroll
            for (int j = 0; j < WN; ++j) fa[j] = *reinterpret_cast<const frag*>(sw + j * (16 * RB) + fo);
#pragma unroll
            for (int i = 0; i < WM; ++i) fb[i] = *reinterpret_cast<const frag*>(sa + i * (16 * RB) + fo);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = ops::mma(fa[j], fb[i], acc[i][j]);
        }
        if (++rbuf == NST) rbuf = 0;
    }
    wait_vmcnt<0>();                                          // the zero-fill pieces issued past the end of K

    // ---- epilogue: lane holds channels c..c+3 of pixel (tile pixel prow + i*16 + fr)
    const int cq = fg * 4;
    float bv[WN][4];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int c = cd0 + crow + j * 16 + cq;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = (bias != nullptr && c < g.Cd) ? bias[c + r] : 0.f;
    }
    {
        int q = m0 + prow + fr;
        const unsigned qq = q < sb.npix ? q : 0;
        const unsigned t2 = qq / (unsigned)sb.Wg;
        int b = (int)(qq - t2 * sb.Wg);
        int n = (int)(t2 / (unsigned)sb.Hg);
        int a = (int)t2 - n * sb.Hg;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const bool live = q < sb.npix;
            const long pix = live ? ((long)n * g.Hd + a * g.ostep + sb.ooff_h) * (long)g.Wd + b * g.ostep + sb.ooff_w : 0;
            store_pixel_blocks<T, WN, ACC>(g, acc[i], bv, dst, pix, live, cd0 + crow, cq, lane);
            q += 16;
            b += 16;
            while (b >= sb.Wg) {
                b -= sb.Wg;
                if (++a == sb.Hg) { a = 0; ++n; }
            }
        }
    }

    // ---- optional BatchNorm batch statistics of the stored (rounded) values, as in k_conv_mfma
    float* const stats = g.stats;
    if (stats != nullptr) {
        __builtin_amdgcn_s_barrier();                         // every wave is past its last fragment read and its DMA drain
        float* sacc = reinterpret_cast<float*>(smem);         // [2][BN]
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

struct RingTile { int bm, bn, nst, bk; };

template <typename T, int RB, int WGM, int WGN, int WM, int WN, int NST>
void launch_ring_t(const RingGeom& g, int nwg, const void* src, const void* wm, const float* bias, void* dst, int accumulate,
                   hipStream_t st) {
    if (accumulate)
        hipLaunchKernelGGL((k_conv_ring<T, RB, WGM, WGN, WM, WN, NST, true>), dim3(nwg), dim3(256), 0, st, g, (const T*)src,
                           (const T*)wm, bias, (T*)dst);
    else
        hipLaunchKernelGGL((k_conv_ring<T, RB, WGM, WGN, WM, WN, NST, false>), dim3(nwg), dim3(256), 0, st, g, (const T*)src,
                           (const T*)wm, bias, (T*)dst);
}

template <typename T>
int launch_ring(const RingGeom& g, const RingTile& t, int nwg, const void* src, const void* wm, const float* bias, void* dst,
                int accumulate, hipStream_t st) {
#define RING_CASE(BK__, BM__, BN__, NST__, WGM_, WGN_, WM_, WN_)                                                      \
    if (t.bk == BK__ && t.bm == BM__ && t.bn == BN__ && t.nst == NST__) {                                             \
        launch_ring_t<T, 2 * BK__, WGM_, WGN_, WM_, WN_, NST__>(g, nwg, src, wm, bias, dst, accumulate, st);          \
        return YOLO_LAUNCH_CHECK();                                                                                   \
    }
#define RING_DEPTHS(BK__, BM__, BN__, WGM_, WGN_, WM_, WN_)                                                           \
    RING_CASE(BK__, BM__, BN__, 2, WGM_, WGN_, WM_, WN_)                                                              \
    RING_CASE(BK__, BM__, BN__, 3, WGM_, WGN_, WM_, WN_)                                                              \
    RING_CASE(BK__, BM__, BN__, 4, WGM_, WGN_, WM_, WN_)
    RING_DEPTHS(64, 128, 128, 2, 2, 4, 4)
    RING_DEPTHS(64, 128, 64, 2, 2, 4, 2)
    RING_DEPTHS(64, 64, 128, 2, 2, 2, 4)
    RING_DEPTHS(64, 64, 64, 2, 2, 2, 2)
    RING_DEPTHS(64, 128, 32, 4, 1, 2, 2)
    RING_DEPTHS(32, 128, 128, 2, 2, 4, 4)
    RING_DEPTHS(32, 128, 64, 2, 2, 4, 2)
    RING_DEPTHS(32, 64, 128, 2, 2, 2, 4)
    RING_DEPTHS(32, 64, 64, 2, 2, 2, 2)
#undef RING_DEPTHS
#undef RING_CASE
    return YOLO_ERR_ARG;
}

}  // namespace

// Shapes the ring kernel takes: `ring` = 1 (yolo_conv_tune_set) forces it wherever it can run, 0 turns it off, -1 (the
// default) uses it where tools/ring_tune.py measured it ahead of the gather kernel on MI355X: maps of 40x40 and below
// with K >= 256 (every step a full cache line per row, half the barriers, the load of step t+1 under the MFMAs of step
// t), and 80x80 maps with K >= 1024.  Larger maps are HBM-bound streams that want the gather kernel's occupancy
// (12 KB of LDS per workgroup instead of 48+).
int ring_conv_eligible(const ConvGeom& g, int dtype, const void* src, const void* wm, const void* dst) {
    const int mode = conv_tune().ring;
    if (mode == 0) return 0;
    if (mode < 0) {
        const long pix = (long)g.N * g.Hd * g.Wd / (g.ostep * g.ostep);       // pixels of one launch (a parity class: a quarter)
        const int steps = g.ntaps * ((g.Cs + 63) / 64);
        const long all_pix = (long)g.N * g.Hd * g.Wd;
        if (!((all_pix <= 60000 && (steps >= 4 || g.ostep == 2)) || (all_pix <= 240000 && steps >= 16))) return 0;
        (void)pix;
    }
    if (dtype != YOLO_BF16 && dtype != YOLO_F16) return 0;
    if (g.Cs % 32 || g.Cs < 64 || g.lds % 8 || g.Cd % 8 || g.ldd % 4) return 0;
    if ((long)g.N * g.Hs * g.Ws * g.lds + (long)(g.Ws + 1) * g.lds >= (1L << 30)) return 0;
    if ((long)g.N * g.Hg * g.Wg + 256 >= (1L << 31) || (long)g.Cd * g.Kpad * 4 >= (1L << 30)) return 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(wm) & 15) || (reinterpret_cast<uintptr_t>(dst) & 7)) return 0;
    for (int t = 0; t < g.ntaps; ++t)
        if (g.dh[t] < -1 || g.dh[t] > 1 || g.dw[t] < -1 || g.dw[t] > 1) return 0;
    return 1;
}

// Tile choice.  Pixel tiles of 128 while that still gives every CU a workgroup, else 64; channel tiles of 128 for wide
// layers while the grid stays at a workgroup per CU, else 64 (32 for <= 32 channels).  K-step and ring depth from
// tools/conv_tune.py (see the table in DESIGN.md).
static RingTile ring_tile(long pix_tiles128, long pix_tiles64, int Cd, int Cs) {
    RingTile t;
    auto wgs = [&](int bm, int bn) { return (bm == 128 ? pix_tiles128 : pix_tiles64) * ((Cd + bn - 1) / bn); };
    t.bn = Cd > 64 ? 128 : (Cd > 32 ? 64 : 32);
    t.bm = 128;
    if (t.bn == 128 && wgs(128, 128) < 256) t.bn = 64;
    if (t.bn != 32 && wgs(128, t.bn) < 256) t.bm = 64;
    const ConvTune& tu = conv_tune();
    if (tu.bn == 32 || tu.bn == 64 || tu.bn == 128) t.bn = tu.bn;
    if (tu.bm == 64 || tu.bm == 128) t.bm = tu.bm;
    t.bk = 64;
    if (tu.bk == 32 || tu.bk == 64) t.bk = tu.bk;
    if (t.bn == 32) { t.bm = 128; t.bk = 64; }                // the 32-channel tile exists for 64-deep steps only
    t.nst = 2;
    if (tu.nst >= 2 && tu.nst <= 4) t.nst = tu.nst;
    return t;
}

static long pix_tiles(const ConvGeom* gs, int n, int bm) {
    long s = 0;
    for (int c = 0; c < n; ++c) s += ((long)gs[c].N * gs[c].Hg * gs[c].Wg + bm - 1) / bm;
    return s;
}

int ring_conv_plan(const ConvGeom* gs, int n) {
    const RingTile t = ring_tile(pix_tiles(gs, n, 128), pix_tiles(gs, n, 64), gs[0].Cd, gs[0].Cs);
    return 3000 + (t.bm == 64 ? 500 : 0) + t.bn;
}

// gs[0..n): sub-problems that share source, destination tensor, channel counts and strides (n = 1, or the four parity
// classes of a stride-2 data gradient; empty classes are skipped); wm_off[c] = element offset of class c's packed
// weight matrix inside wm, wm_elems = size of the whole buffer
int ring_conv_launch(const ConvGeom* gs, int n, const long* wm_off, long wm_elems, const void* src, const void* wm,
                     const float* bias, void* dst, int accumulate, int dtype, hipStream_t st) {
    if (n < 1 || n > 4 || wm_elems >= (1L << 30)) return YOLO_ERR_ARG;
    const ConvGeom& g0 = gs[0];
    const RingTile t = ring_tile(pix_tiles(gs, n, 128), pix_tiles(gs, n, 64), g0.Cd, g0.Cs);
    RingGeom d;
    d.N = g0.N; d.Hs = g0.Hs; d.Ws = g0.Ws; d.Cs = g0.Cs; d.lds = g0.lds; d.Hd = g0.Hd; d.Wd = g0.Wd; d.Cd = g0.Cd; d.ldd = g0.ldd;
    d.ostep = g0.ostep; d.sstride = g0.sstride; d.stats = g0.stats;
    d.acc2 = g0.acc2; d.ld2 = g0.ld2;
    d.act = g0.act; d.res = g0.res; d.ldr = g0.ldr;
    d.wide = to_dev(g0).wide;
    d.spt = (g0.Cs + t.bk - 1) / t.bk;
    d.ntile_n = (g0.Cd + t.bn - 1) / t.bn;
    d.wm_elems = (int)wm_elems;
    d.nsub = 0;
    int tiles = 0;
    for (int c = 0; c < n; ++c) {
        const ConvGeom& g = gs[c];
        const long npix = (long)g.N * g.Hg * g.Wg;
        if (npix == 0) continue;
        RingSub& s = d.sub[d.nsub++];
        s.Hg = g.Hg; s.Wg = g.Wg; s.ooff_h = g.ooff_h; s.ooff_w = g.ooff_w; s.ntaps = g.ntaps; s.KT = g.ntaps * d.spt; s.Kpad = g.Kpad;
        s.wm_off = (int)wm_off[c]; s.tile0 = tiles; s.npix = (int)npix;
        s.dh_pack = s.dw_pack = 0;
        for (int k = 0; k < g.ntaps; ++k) {
            s.dh_pack |= (unsigned)(g.dh[k] + 1) << (2 * k);
            s.dw_pack |= (unsigned)(g.dw[k] + 1) << (2 * k);
        }
        tiles += (int)((npix + t.bm - 1) / t.bm);
    }
    if (d.nsub == 0) return YOLO_OK;
    for (int c = d.nsub; c < 4; ++c) d.sub[c] = d.sub[0];
    const int nwg = tiles * d.ntile_n;
    if (dtype == YOLO_BF16) return launch_ring<bf16_t>(d, t, nwg, src, wm, bias, dst, accumulate, st);
    return launch_ring<f16_t>(d, t, nwg, src, wm, bias, dst, accumulate, st);
}
