// Weight gradient of a dense conv on the matrix cores, second design.
//   dW[co][tap][ci] = sum over output pixels p of dY[p][co] * X[p*s + tap - pad][ci]
// The reduction index is the pixel.  One workgroup owns a (32*TCO x 32*TCI) block of (co, ci) for ALL
// k*k taps and walks a slab of 4x8 output-pixel patches: per patch it stages dY (32 pixels) and the X
// patch with its halo ((3s+k) x (7s+k) pixels) ONCE, in their natural [pixel][channel] layout, and
// feeds the MFMAs through ds_read_b64_tr_b16 -- the hardware transposing read turns "pixel-major"
// into the K-major fragment both operands need, at any pixel offset, so the 9 taps are 9 address
// offsets into the same patch (a [channel][pixel] LDS image would need a misaligned read per tap).
// MFMA k index j <-> patch pixel (row j/8, col j%8): lane group g = patch row g.
// Slabs are combined with fp32 atomics into the packed gradient matrix (64-byte runs along ci).
#include <cstdio>
#include <cstdlib>
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct mm;
template <> struct mm<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct mm<f16_t> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// LDS image.  ds_read_b64_tr_b16 is served in two groups of 32 lanes, bank = (byte address / 4) mod 64
// (MI355X_MICROARCH.md, LDS).  In a transposing fragment read the 32 lanes of a group cover EIGHT pixel rows x 32 bytes (one
// 16-channel block): conflict-free iff the eight rows start on eight different multiples of 8 banks.  Two choices make that
// hold for every read of the kernel:
//  * row stride = 32*t + 16 elements = 8 * (2t + 1) banks, an ODD multiple of 8: rows r .. r+7 then start on 8 different
//    multiples of 8 banks (round 2's strides -- 16, 36, 48, 72 banks -- put them two to four deep: 38 % of the kernel's LDS
//    cycles were conflict cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE);
//  * the eight rows of a read are made two runs of four rows 12 apart (12 = 4 mod 8) or one run of eight: dY pixel j of the
//    4 x 8 patch sits in row (j&3) + 4*((j>>3)&1) + 8*((j>>2)&1) + 16*(j>>4) (a read takes pixels 8g+q and 8(g+1)+q, q = 0..3);
//    the X patch is stored 12 pixels wide (PWL), and for stride 2 as four parity planes (row parity, column parity) so
//    that the pixels two apart which a stride-2 tap reads are neighbours in their plane.
constexpr int ldc_of(int t) { return t * 32 + 16; }
constexpr int PWL = 12;                                      // LDS width (pixels) of the X patch / of one parity plane

struct WgArgs {
    int N, H, W, Cin, ldx, OH, OW, Cout, ldy, Kpad;
    int pbh, pbw;            // patches per image along h / w
    long npatch, per_slab;
    int cot, cit;            // (co, ci) tiles
    int xcd;                 // 1: XCD-aware order of the work items
};

// bijective XCD-aware remap (workgroup i runs on XCD i % 8): workgroups that share an XCD get a contiguous range of
// work items, so the tiles of one slab -- which read the same X / dY rows -- sit behind one L2
__device__ __forceinline__ int xcd_chunk(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// two transposing reads -> one MFMA fragment: element j = LDS[(row0 + (j&3) + 4*(j>>2)*rstep ... )]
template <typename T>
__device__ __forceinline__ typename mm<T>::frag tr_frag(const T* p_lo, const T* p_hi) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_lo);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_hi);
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename mm<T>::frag, both);
}

// PF = patches in flight per workgroup (registers: 12-24 per patch).  PF = 4 was built for the layers with at most one
// workgroup per CU (every 20x20 / 40x40 map) on the assumption that they are bound by the round trip of their patch loads.
// Measured (tools/wg_scan.py, profiles/r3_wgrad_scan.log): they are not -- the marginal cost of a patch is 0.4-0.6 us
// (3x3) whatever the depth, and ~20 us of every small layer is FIXED: two launches, the partial-matrix store of every
// workgroup and the reduce pass.  PF = 4 is 5-15 % slower (254 registers, one full drain of the ring per loop trip at the
// loop header, where the compiler's waitcnt analysis gives up) and is kept only as a tested variant (yolo_wgrad_tune_pf).
template <typename T, int KS, int S, int TCO, int TCI, int PF>
__global__ __launch_bounds__(256) void k_wgrad2(WgArgs a, const T* __restrict__ x, const T* __restrict__ dy,
                                                float* __restrict__ dwp) {
    constexpr int PAD = KS / 2;
    constexpr int PH = 3 * S + KS, PW = 7 * S + KS;          // X patch (with halo) for 4 x 8 outputs
    constexpr int LDY = ldc_of(TCO), LDX = ldc_of(TCI);
    constexpr int SUB = ((PH + 1) / 2) * PWL;                 // stride 2: LDS rows of one parity plane
    constexpr int XROWS = S == 1 ? PH * PWL : 4 * SUB;        // LDS rows of the X image
    constexpr int CPY = TCO * 4, CPX = TCI * 4;               // 16-byte chunks per pixel (dY / X tile)
    constexpr int YCH = 32 * CPY, YR = (YCH + 255) / 256;     // dY chunks per patch / per thread
    constexpr int XCH = PH * PW * CPX;                        // 16-byte chunks in the X patch
    constexpr int XR = (XCH + 255) / 256;                     // chunks per thread
    constexpr int NT = KS * KS;
    static_assert(PW <= PWL || S == 2, "X patch wider than its LDS image");
    static_assert(S == 1 || (PW + 1) / 2 <= PWL, "parity plane wider than its LDS image");
    __shared__ __attribute__((aligned(16))) T ys[32 * LDY];
    __shared__ __attribute__((aligned(16))) T xs[XROWS * LDX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wi = wave & 1;
    // block tile = (2 waves x TCO x 16) co  x  (2 waves x TCI x 16) ci
    const int unit = a.xcd ? xcd_chunk(blockIdx.x, gridDim.x) : (int)blockIdx.x, ntile = a.cot * a.cit;
    const int slab = unit / ntile, tile = unit - slab * ntile;
    const int bco = (tile % a.cot) * (32 * TCO), bci = (tile / a.cot) * (32 * TCI);
    long p_begin = (long)slab * a.per_slab, p_end = p_begin + a.per_slab;
    if (p_end > a.npatch) p_end = a.npatch;

    f32x4 acc[NT][TCO][TCI];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TCI; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- loader.  Both tensors are read through buffer descriptors: a thread's chunk has a FIXED byte offset
    // inside the patch (voffset), the patch origin is one scalar (soffset), and a chunk outside the image or
    // the channel range gets a voffset past the descriptor's range, which reads as zero.  The descriptor of X
    // starts one row and one pixel before the tensor so the origin of a border patch is never negative.
    // Patch coordinates (n, bh, bw) advance by carrying -- no division in the loop.  (host: sizes < 2^30 elements)
    const int xshift = (a.W + 1) * a.ldx;
    const __amdgpu_buffer_rsrc_t rsx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x) - xshift, 0, (a.N * a.H * a.W * a.ldx + xshift) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy), 0, a.N * a.OH * a.OW * a.ldy * 2, 0x00020000);
    int yrow[YR], ycol[YR], yvoff[YR], ylds[YR];             // dY chunk: pixel (row, col) of the 4x8 patch + channel chunk
#pragma unroll
    for (int r = 0; r < YR; ++r) {
        const int id = tid + 256 * r;
        const int j = id / CPY, ch = (id - j * CPY) * 8;
        yrow[r] = j >> 3;
        ycol[r] = j & 7;
        yvoff[r] = (id < YCH && bco + ch < a.Cout) ? ((yrow[r] * a.OW + ycol[r]) * a.ldy + bco + ch) * 2 : (int)0x80000000;
        ylds[r] = id < YCH ? ((j & 3) + 4 * ((j >> 3) & 1) + 8 * ((j >> 2) & 1) + 16 * (j >> 4)) * LDY + ch : -1;
    }
    int xpr[XR], xpc[XR], xvoff[XR], xlds[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int id = tid + 256 * r;
        const int px = id / CPX, ch = (id - px * CPX) * 8;
        xpr[r] = px / PW;
        xpc[r] = px - xpr[r] * PW;
        xvoff[r] = (id < XCH && bci + ch < a.Cin) ? ((xpr[r] * a.W + xpc[r]) * a.ldx + bci + ch) * 2 : (int)0x80000000;
        const int row = S == 1 ? xpr[r] * PWL + xpc[r]
                               : ((xpr[r] & 1) * 2 + (xpc[r] & 1)) * SUB + (xpr[r] >> 1) * PWL + (xpc[r] >> 1);
        xlds[r] = id < XCH ? row * LDX + ch : -1;
    }
    int pn, pbh, pbw;                                        // coordinates of the patch gload fetches next
    {
        const int per_img = a.pbh * a.pbw;
        pn = (int)(p_begin / per_img);
        const int rem = (int)(p_begin - (long)pn * per_img);
        pbh = rem / a.pbw;
        pbw = rem - pbh * a.pbw;
    }
    uint4 ry[PF][YR], rx[PF][XR];
    // `live` false (a slot past the end of the slab): every chunk gets the out-of-range offset -- no memory traffic, zeros
    // in the registers -- so that each step issues the SAME number of loads and the compiler's s_waitcnt vmcnt(N) in front
    // of a slot's LDS stores can count the younger slots' loads instead of waiting for all of them (with the loads under a
    // branch it emitted vmcnt(2..0): the whole ring drained every step, 10-20 % SLOWER than one patch in flight)
    auto gload = [&](uint4 (&ry_)[YR], uint4 (&rx_)[XR], bool live) {
        const int oh0 = pbh * 4, ow0 = pbw * 8;
        const int ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
        const int ysoff = live ? ((pn * a.OH + oh0) * a.OW + ow0) * a.ldy * 2 : 0;
#pragma unroll
        for (int r = 0; r < YR; ++r) {
            const bool ok = live && oh0 + yrow[r] < a.OH && ow0 + ycol[r] < a.OW;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsy, ok ? yvoff[r] : (int)0x80000000, ysoff, 0);
            ry_[r] = make_uint4(v.x, v.y, v.z, v.w);
        }
        const int xsoff = live ? (((pn * a.H + ih0) * a.W + iw0) * a.ldx + xshift) * 2 : 0;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const bool ok = live && (unsigned)(ih0 + xpr[r]) < (unsigned)a.H && (unsigned)(iw0 + xpc[r]) < (unsigned)a.W;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? xvoff[r] : (int)0x80000000, xsoff, 0);
            rx_[r] = make_uint4(v.x, v.y, v.z, v.w);
        }
        if (++pbw == a.pbw) {
            pbw = 0;
            if (++pbh == a.pbh) { pbh = 0; ++pn; }
        }
    };
    auto lstore = [&](const uint4 (&ry_)[YR], const uint4 (&rx_)[XR]) {
#pragma unroll
        for (int r = 0; r < YR; ++r)
            if (ylds[r] >= 0) *reinterpret_cast<uint4*>(ys + ylds[r]) = ry_[r];
#pragma unroll
        for (int r = 0; r < XR; ++r)
            if (xlds[r] >= 0) *reinterpret_cast<uint4*>(xs + xlds[r]) = rx_[r];
    };

    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, c4 = 4 * (i16 & 3);
    // dY^T fragment rows: pixels 8g+q (lo) and 8g+4+q (hi) -> LDS rows q + 4*(g&1) + 16*(g>>1) and that + 8
    const T* ya = ys + (q + 4 * (g & 1) + 16 * (g >> 1)) * LDY + wc * TCO * 16 + c4;
    auto compute = [&]() {
        // (Issuing all 40 fragment reads of a patch before its first MFMA -- counted lgkmcnt waits, sched_barrier between the
        // phases -- was measured 4 % SLOWER over the layers of preset s than this interleaved order.)
        // A fragments (dY^T): rows = co, k = patch pixel 8g + j
        typename mm<T>::frag fa[TCO];
#pragma unroll
        for (int i = 0; i < TCO; ++i) fa[i] = tr_frag<T>(ya + i * 16, ya + i * 16 + 8 * LDY);
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                // output pixel (row g, col q | q+4) reads X patch pixel (g*S + kh, col*S + kw)
                const int row = S == 1 ? (g + kh) * PWL + q + kw
                                       : ((kh & 1) * 2 + (kw & 1)) * SUB + (g + (kh >> 1)) * PWL + q + (kw >> 1);
                const T* base = xs + row * LDX + wi * TCI * 16 + c4;
#pragma unroll
                for (int j = 0; j < TCI; ++j) {
                    typename mm<T>::frag fb = tr_frag<T>(base + j * 16, base + j * 16 + 4 * LDX);
#pragma unroll
                    for (int i = 0; i < TCO; ++i) acc[kh * KS + kw][i][j] = mm<T>::mma(fa[i], fb, acc[kh * KS + kw][i][j]);
                }
            }
    };
#pragma unroll
    for (int s2 = 0; s2 < PF; ++s2) gload(ry[s2], rx[s2], p_begin + s2 < p_end);
    for (long pi = p_begin; pi < p_end; pi += PF) {
#pragma unroll
        for (int s2 = 0; s2 < PF; ++s2) {
            if (pi + s2 >= p_end) break;
            __syncthreads();
            lstore(ry[s2], rx[s2]);
            __syncthreads();
            gload(ry[s2], rx[s2], pi + s2 + PF < p_end);
            compute();
        }
    }
    // Partial sums leave as a raw register image: part[slab][tile][wave][tap][i][j][lane] is the lane's float4 accumulator
    // (rows co = g*4 + r, column ci = lane & 15 of the 16x16 block (i, j)) -- one fully coalesced 1 KB store per accumulator
    // block instead of four 64-byte runs per register (the [Cout][Kpad] image cost 13 of a 20x20 128->128 layer's 21 us);
    // k_wgrad_reduce undoes the permutation while it sums the slabs.  Plain stores: an atomic flush into one shared matrix ran
    // at ~250 G adds/s and dominated the small layers.
    float4* part4 = reinterpret_cast<float4*>(dwp) + ((((long)slab * ntile + tile) * 4 + wave) * (NT * TCO * TCI)) * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TCI; ++j) {
                const f32x4 v = acc[t][i][j];
                part4[((t * TCO + i) * TCI + j) * 64] = make_float4(v[0], v[1], v[2], v[3]);
            }
}

// sum of the slabs' partial images -> OIHW gradient in the parameter's dtype (replaces memset + unpack).
// A thread owns one float4 of the register image (four co rows of one ci column of one tap: see k_wgrad2's epilogue): one
// 16-byte load per slab, all lanes of a wave on consecutive 16-byte pieces; SL lanes share a group and stride over the slabs,
// then combine through LDS; the four sums go to their places in [Cout][Cin][k][k].
template <typename TO, int SL>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int nslab, int Cout, int Cin, int NT,
                                                      int tco, int tci, int cot, long groups, TO* __restrict__ dw) {
    constexpr int EL = 256 / SL;                              // float4 groups per workgroup
    __shared__ float4 red[SL][EL + 1];
    const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
    const long gi = (long)blockIdx.x * EL + el;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gi < groups) {
        const float4* p = reinterpret_cast<const float4*>(part) + gi;
#pragma unroll 4
        for (int s2 = sl; s2 < nslab; s2 += SL) {
            const float4 v = p[s2 * groups];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    if (SL > 1) {
        red[sl][el] = a;
        __syncthreads();
        if (sl != 0) return;
#pragma unroll
        for (int s2 = 1; s2 < SL; ++s2) {
            const float4 v = red[s2][el];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    if (gi < groups) {
        const int lane = (int)(gi & 63);
        long rest = gi >> 6;
        const int j = (int)(rest % tci); rest /= tci;
        const int i = (int)(rest % tco); rest /= tco;
        const int t = (int)(rest % NT); rest /= NT;
        const int wave = (int)(rest & 3);
        const int tile = (int)(rest >> 2);
        const int co = (tile % cot) * (32 * tco) + ((wave >> 1) * tco + i) * 16 + (lane >> 4) * 4;
        const int ci = (tile / cot) * (32 * tci) + ((wave & 1) * tci + j) * 16 + (lane & 15);
        if (ci < Cin) {
            TO* o = dw + ((long)co * Cin + ci) * NT + t;
            const long rs = (long)Cin * NT;
            if (co < Cout) o[0] = (TO)a.x;
            if (co + 1 < Cout) o[rs] = (TO)a.y;
            if (co + 2 < Cout) o[2 * rs] = (TO)a.z;
            if (co + 3 < Cout) o[3 * rs] = (TO)a.w;
        }
    }
}

struct WgPlan { int to, ti, cot, cit, nslab, pf; long per_slab; };

template <typename TO>
void launch_reduce(const float* part, const WgPlan& p, int Cout, int Cin, int NT, void* dw, hipStream_t st) {
    const long groups = (long)p.cot * p.cit * 4 * NT * p.to * p.ti * 64;
    if (p.nslab >= 32)
        hipLaunchKernelGGL((k_wgrad_reduce<TO, 8>), dim3((unsigned)((groups + 31) / 32)), dim3(256), 0, st, part, p.nslab, Cout, Cin, NT, p.to, p.ti, p.cot, groups, (TO*)dw);
    else if (p.nslab >= 4)
        hipLaunchKernelGGL((k_wgrad_reduce<TO, 4>), dim3((unsigned)((groups + 63) / 64)), dim3(256), 0, st, part, p.nslab, Cout, Cin, NT, p.to, p.ti, p.cot, groups, (TO*)dw);
    else
        hipLaunchKernelGGL((k_wgrad_reduce<TO, 1>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, part, p.nslab, Cout, Cin, NT, p.to, p.ti, p.cot, groups, (TO*)dw);
}



// Plan overrides (tile, workgroup count), 0 = automatic.  Read ONCE per process from YOLO_WG_TUNE ("to,ti,blocks,min_per")
// and YOLO_WG_BLOCKS; tools/wg_tune.py and the plan-forcing parity tests change them through yolo_wgrad_tune_set.
struct WgTune { int to, ti, blocks, min_per; long all_blocks; int pf; };
WgTune& wg_tune() {
    static WgTune t = [] {
        WgTune v{0, 0, 0, 0, 0, 0};
        if (const char* e = getenv("YOLO_WG_TUNE")) sscanf(e, "%d,%d,%d,%d", &v.to, &v.ti, &v.blocks, &v.min_per);
        if (const char* e2 = getenv("YOLO_WG_BLOCKS")) v.all_blocks = atol(e2);
        if (const char* e3 = getenv("YOLO_WG_PF")) v.pf = atoi(e3);         // 1 / 4: force the prefetch depth (A/B runs)
        return v;
    }();
    return t;
}

// tile and slab choice (shared by the workspace query and the launch)
WgPlan make_plan(const WgArgs& a, int k) {
    WgPlan p;
    // Tuned on MI355X (tools/wg_tune.py, preset-s layer shapes at 32 images):
    //  1x1: the accumulators of a (32*TCO x 32*TCI) tile are TCO*TCI*4 registers, so a workgroup owns up to 128 x 128
    //       of (co, ci) and X / dY are each read once for layers up to 128 channels; the widest tile won everywhere.
    //  3x3: 9 taps x 64 x 64 is 144 accumulator registers; for Cin <= 64 the 64 x 32 tile (72) runs at twice the
    //       occupancy and wins although dY is then read twice.
    //  ~2 workgroups per CU overall: more slabs only add partial-matrix traffic.
    if (k == 1) {
        p.to = a.Cout > 96 ? 4 : a.Cout > 64 ? 3 : a.Cout > 32 ? 2 : 1;
        p.ti = a.Cin > 96 ? 4 : a.Cin > 64 ? 3 : a.Cin > 32 ? 2 : 1;
    } else {
        p.to = a.Cout > 32 ? 2 : 1;
        p.ti = a.Cin > 64 ? 2 : 1;
    }
    // workgroup count (= slabs x tiles): every workgroup ends by storing its tile's partial sums, so more slabs
    // mean more partial-matrix traffic (PMC: ~4 GB per step); fewer leave CUs idle.  From tools/wg_tune.py:
    const int cot0 = (a.Cout + 32 * p.to - 1) / (32 * p.to), cit0 = (a.Cin + 32 * p.ti - 1) / (32 * p.ti);
    const long data_bytes = ((long)a.N * a.H * a.W * a.Cin + (long)a.N * a.OH * a.OW * a.Cout) * 2;
    long want_blocks;
    const WgTune& tu = wg_tune();                             // overrides: tuning runs and plan-forcing tests only
    if (tu.all_blocks > 0) {                                  // one count for every layer
        want_blocks = tu.all_blocks;
    } else if (k == 1) {
        if (p.to * p.ti == 1) want_blocks = 2048;                                  // 32 x 32 tiles: 4 KB partials
        else want_blocks = (cot0 * cit0 == 1 && data_bytes < (150L << 20)) ? 256 : 512;
    } else {
        want_blocks = p.ti == 1 ? 512 : 256;                                       // 64 x 32 x 9 vs 64 x 64 x 9 tiles
    }
    // small layers: no more partial-matrix bytes than `scale` x the operand bytes (the partials of a 20x20 layer were
    // 5-10x its operands at the counts above: ~65 MB written and read back per layer whatever its size), but never fewer
    // than `floor` workgroups.  In-step A/B (same box): off 11.75 ms, 0.5/128 11.51, 0.35-0.7 / 96-192 11.55-11.58,
    // 0.25/64 11.99, 0.125/64 12.77 (too few workgroups: the side stream becomes the critical path)
    {
        static const double scale = [] { const char* e = getenv("YOLO_WG_SCALE"); return e ? atof(e) : 0.5; }();
        if (scale > 0.0 && tu.all_blocks <= 0 && tu.blocks <= 0) {
            const long tile_bytes = (long)k * k * (32 * p.to) * (32 * p.ti) * 4;
            long cap = (long)(scale * (double)data_bytes / (double)tile_bytes);
            static const long floor_wg = [] { const char* e = getenv("YOLO_WG_FLOOR"); return e ? atol(e) : 128L; }();
            if (cap < floor_wg) cap = floor_wg;
            if (want_blocks > cap) want_blocks = cap;
        }
    }
    long min_per = 2;
    if (tu.to > 0 && (k == 1 || tu.to <= 2)) p.to = tu.to;
    if (tu.ti > 0 && (k == 1 || tu.ti <= 2)) p.ti = tu.ti;
    if (tu.blocks > 0) want_blocks = tu.blocks;
    if (tu.min_per > 0) min_per = tu.min_per;
    p.cot = (a.Cout + 32 * p.to - 1) / (32 * p.to);
    p.cit = (a.Cin + 32 * p.ti - 1) / (32 * p.ti);
    // ~2 workgroups per CU for 3x3 (144 accumulator registers), ~4 for 1x1; every workgroup ends by storing its
    // KS*KS*(32*TCO)*(32*TCI) partial sums, so it should see at least a few patches first
    long want = want_blocks / ((long)p.cot * p.cit);
    if (want > a.npatch / min_per) want = a.npatch / min_per;
    if (want < 1) want = 1;
    p.per_slab = (a.npatch + want - 1) / want;
    p.nslab = (int)((a.npatch + p.per_slab - 1) / p.per_slab);
    p.pf = 1;                                                 // see k_wgrad2: four patches in flight measured slower
    if (tu.pf == 1 || tu.pf == 4) p.pf = tu.pf;
    return p;
}

template <typename T, int KS, int S, int TCO, int TCI>
void launch(const WgArgs& a, const WgPlan& p, const void* x, const void* dy, float* part, hipStream_t st) {
    WgArgs b = a;
    b.per_slab = p.per_slab;
    b.cot = p.cot; b.cit = p.cit;
    static const int xcd = [] { const char* e = getenv("YOLO_WG_XCD"); return e ? atoi(e) : 1; }();
    b.xcd = xcd;
    const dim3 grid((unsigned)(p.cot * p.cit * p.nslab));
    if (p.pf == 4)
        hipLaunchKernelGGL((k_wgrad2<T, KS, S, TCO, TCI, 4>), grid, dim3(256), 0, st, b, (const T*)x, (const T*)dy, part);
    else
        hipLaunchKernelGGL((k_wgrad2<T, KS, S, TCO, TCI, 1>), grid, dim3(256), 0, st, b, (const T*)x, (const T*)dy, part);
}

template <typename T, int KS, int S>
void launch_tiles(const WgArgs& a, const WgPlan& p, const void* x, const void* dy, float* part, hipStream_t st) {
#define WG_CASE(TO_, TI_) if (p.to == TO_ && p.ti == TI_) return launch<T, KS, S, TO_, TI_>(a, p, x, dy, part, st)
    if constexpr (KS == 1) {
        WG_CASE(1, 1); WG_CASE(1, 2); WG_CASE(1, 3); WG_CASE(1, 4);
        WG_CASE(2, 1); WG_CASE(2, 2); WG_CASE(2, 3); WG_CASE(2, 4);
        WG_CASE(3, 1); WG_CASE(3, 2); WG_CASE(3, 3); WG_CASE(3, 4);
        WG_CASE(4, 1); WG_CASE(4, 2); WG_CASE(4, 3); WG_CASE(4, 4);
    } else {
        WG_CASE(1, 1); WG_CASE(1, 2); WG_CASE(2, 1); WG_CASE(2, 2);
    }
#undef WG_CASE
}

template <typename T>
int launch_ks(const WgArgs& a, const WgPlan& p, int k, int stride, const void* x, const void* dy, float* part,
              hipStream_t st) {
    if (k == 1 && stride == 1) launch_tiles<T, 1, 1>(a, p, x, dy, part, st);
    else if (k == 3 && stride == 1) launch_tiles<T, 3, 1>(a, p, x, dy, part, st);
    else if (k == 3 && stride == 2) launch_tiles<T, 3, 2>(a, p, x, dy, part, st);
    else return YOLO_ERR_ARG;
    return YOLO_LAUNCH_CHECK();
}

WgArgs make_args(int ldx, int ldy, int Kpad, int N, int H, int W, int Cin, int OH, int OW, int Cout) {
    WgArgs a;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.OH = OH; a.OW = OW; a.Cout = Cout; a.ldy = ldy; a.Kpad = Kpad;
    a.pbh = (OH + 3) / 4; a.pbw = (OW + 7) / 8;
    a.npatch = (long)N * a.pbh * a.pbw;
    a.per_slab = 0;
    return a;
}

}  // namespace

extern "C" int yolo_wgrad_tune_set(int to, int ti, int blocks, int min_per) {
    WgTune& t = wg_tune();
    t.to = to; t.ti = ti; t.blocks = blocks; t.min_per = min_per;
    return YOLO_OK;
}

// test / tuning override of the patches-in-flight variant: 0 = automatic, 1 or 4
extern "C" int yolo_wgrad_tune_pf(int pf) {
    if (pf != 0 && pf != 1 && pf != 4) return YOLO_ERR_ARG;
    wg_tune().pf = pf;
    return YOLO_OK;
}

// fp32 elements of partial-sum scratch the launch below writes: one register image of every (co, ci) tile per slab
long mfma_wgrad2_ws_elems(int Kpad, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k) {
    WgArgs a = make_args(0, 0, Kpad, N, H, W, Cin, OH, OW, Cout);
    if (a.npatch == 0) return (long)Cout * Kpad;
    const WgPlan p = make_plan(a, k);
    return (long)p.nslab * p.cot * p.cit * k * k * (32L * p.to) * (32L * p.ti);
}

long mfma_wgrad2_plan(int Kpad, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k) {
    WgArgs a = make_args(0, 0, Kpad, N, H, W, Cin, OH, OW, Cout);
    if (a.npatch == 0) return 0;
    const WgPlan p = make_plan(a, k);
    return p.pf * 10000000L + p.to * 1000000L + p.ti * 100000L + p.nslab;
}

// part (fp32 scratch of mfma_wgrad2_ws_elems elements, need not be zeroed) <- per-slab partial gradients, then
// dw_oihw[Cout][Cin][k][k] (dw_dtype) <- their sum
int mfma_wgrad2_launch(const void* x, int ldx, const void* dy, int ldy, float* part, void* dw_oihw, int dw_dtype, int Kpad,
                       int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, hipStream_t st) {
    WgArgs a = make_args(ldx, ldy, Kpad, N, H, W, Cin, OH, OW, Cout);
    if (a.npatch == 0) return YOLO_ERR_ARG;
    const WgPlan p = make_plan(a, k);
    int rc;
    if (dtype == YOLO_BF16) rc = launch_ks<bf16_t>(a, p, k, stride, x, dy, part, st);
    else if (dtype == YOLO_F16) rc = launch_ks<f16_t>(a, p, k, stride, x, dy, part, st);
    else return YOLO_ERR_DTYPE;
    if (rc) return rc;
    if (dw_dtype == YOLO_F32) launch_reduce<float>(part, p, Cout, Cin, k * k, dw_oihw, st);
    else if (dw_dtype == YOLO_BF16) launch_reduce<bf16_t>(part, p, Cout, Cin, k * k, dw_oihw, st);
    else if (dw_dtype == YOLO_F16) launch_reduce<f16_t>(part, p, Cout, Cin, k * k, dw_oihw, st);
    else return YOLO_ERR_DTYPE;
    return YOLO_LAUNCH_CHECK();
}
