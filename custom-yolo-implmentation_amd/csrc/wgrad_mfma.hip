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

// LDS row (one pixel) of a T*32-channel tile: pad so that the four pixel rows of a transposing read land on
// different banks (row stride in banks mod 64: 32 ch -> 16, 64 -> 36, 96 -> 48, 128 -> 8)
constexpr int ldc_of(int t) { return t == 2 ? 72 : t == 4 ? 144 : t * 32; }

struct WgArgs {
    int N, H, W, Cin, ldx, OH, OW, Cout, ldy, Kpad;
    int pbh, pbw;            // patches per image along h / w
    long npatch, per_slab;
};

// two transposing reads -> one MFMA fragment: element j = LDS[(row0 + (j&3) + 4*(j>>2)*rstep ... )]
template <typename T>
__device__ __forceinline__ typename mm<T>::frag tr_frag(const T* p_lo, const T* p_hi) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_lo);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_hi);
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename mm<T>::frag, both);
}

template <typename T, int KS, int S, int TCO, int TCI>
__global__ __launch_bounds__(256) void k_wgrad2(WgArgs a, const T* __restrict__ x, const T* __restrict__ dy,
                                                float* __restrict__ dwp) {
    constexpr int PAD = KS / 2;
    constexpr int PH = 3 * S + KS, PW = 7 * S + KS;          // X patch (with halo) for 4 x 8 outputs
    constexpr int LDY = ldc_of(TCO), LDX = ldc_of(TCI);
    constexpr int CPY = TCO * 4, CPX = TCI * 4;               // 16-byte chunks per pixel (dY / X tile)
    constexpr int YCH = 32 * CPY, YR = (YCH + 255) / 256;     // dY chunks per patch / per thread
    constexpr int XCH = PH * PW * CPX;                        // 16-byte chunks in the X patch
    constexpr int XR = (XCH + 255) / 256;                     // chunks per thread
    constexpr int NT = KS * KS;
    __shared__ __attribute__((aligned(16))) T ys[32 * LDY];
    __shared__ __attribute__((aligned(16))) T xs[PH * PW * LDX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wi = wave & 1;
    // block tile = (2 waves x TCO x 16) co  x  (2 waves x TCI x 16) ci
    const int bco = blockIdx.x * (32 * TCO), bci = blockIdx.y * (32 * TCI);
    long p_begin = (long)blockIdx.z * a.per_slab, p_end = p_begin + a.per_slab;
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
    int yrow[YR], ycol[YR], yvoff[YR];                       // dY chunk: pixel (row, col) of the 4x8 patch + channel chunk
#pragma unroll
    for (int r = 0; r < YR; ++r) {
        const int id = tid + 256 * r;
        const int j = id / CPY, ch = (id - j * CPY) * 8;
        yrow[r] = j >> 3;
        ycol[r] = j & 7;
        yvoff[r] = (id < YCH && bco + ch < a.Cout) ? ((yrow[r] * a.OW + ycol[r]) * a.ldy + bco + ch) * 2 : (int)0x80000000;
    }
    int xpr[XR], xpc[XR], xvoff[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int id = tid + 256 * r;
        const int px = id / CPX, ch = (id - px * CPX) * 8;
        xpr[r] = px / PW;
        xpc[r] = px - xpr[r] * PW;
        xvoff[r] = (id < XCH && bci + ch < a.Cin) ? ((xpr[r] * a.W + xpc[r]) * a.ldx + bci + ch) * 2 : (int)0x80000000;
    }
    int pn, pbh, pbw;                                        // coordinates of the patch gload fetches next
    {
        const int per_img = a.pbh * a.pbw;
        pn = (int)(p_begin / per_img);
        const int rem = (int)(p_begin - (long)pn * per_img);
        pbh = rem / a.pbw;
        pbw = rem - pbh * a.pbw;
    }
    uint4 ry[YR], rx[XR];
    auto gload = [&]() {
        const int oh0 = pbh * 4, ow0 = pbw * 8;
        const int ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
        const int ysoff = ((pn * a.OH + oh0) * a.OW + ow0) * a.ldy * 2;
#pragma unroll
        for (int r = 0; r < YR; ++r) {
            const bool ok = oh0 + yrow[r] < a.OH && ow0 + ycol[r] < a.OW;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsy, ok ? yvoff[r] : (int)0x80000000, ysoff, 0);
            ry[r] = make_uint4(v.x, v.y, v.z, v.w);
        }
        const int xsoff = (((pn * a.H + ih0) * a.W + iw0) * a.ldx + xshift) * 2;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const bool ok = (unsigned)(ih0 + xpr[r]) < (unsigned)a.H && (unsigned)(iw0 + xpc[r]) < (unsigned)a.W;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? xvoff[r] : (int)0x80000000, xsoff, 0);
            rx[r] = make_uint4(v.x, v.y, v.z, v.w);
        }
        if (++pbw == a.pbw) {
            pbw = 0;
            if (++pbh == a.pbh) { pbh = 0; ++pn; }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int r = 0; r < YR; ++r) {
            const int id = tid + 256 * r;
            if (id < YCH) *reinterpret_cast<uint4*>(ys + (id / CPY) * LDY + (id % CPY) * 8) = ry[r];
        }
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int id = tid + 256 * r;
            if (id < XCH) *reinterpret_cast<uint4*>(xs + (id / CPX) * LDX + (id % CPX) * 8) = rx[r];
        }
    };

    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, c4 = 4 * (i16 & 3);
    if (p_begin < p_end) gload();
    for (long pi = p_begin; pi < p_end; ++pi) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (pi + 1 < p_end) gload();
        // A fragments (dY^T): rows = co, k = patch pixel 8g + j  ->  LDS pixel rows 8g+q and 8g+4+q
        typename mm<T>::frag fa[TCO];
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const T* p = ys + (8 * g + q) * LDY + (wc * TCO + i) * 16 + c4;
            fa[i] = tr_frag<T>(p, p + 4 * LDY);
        }
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                // output pixel (row g, col q | q+4) reads X patch pixel (g*S + kh, col*S + kw)
                const T* base = xs + ((g * S + kh) * PW + q * S + kw) * LDX + c4;
#pragma unroll
                for (int j = 0; j < TCI; ++j) {
                    const T* p = base + (wi * TCI + j) * 16;
                    typename mm<T>::frag fb = tr_frag<T>(p, p + 4 * S * LDX);
#pragma unroll
                    for (int i = 0; i < TCO; ++i) acc[kh * KS + kw][i][j] = mm<T>::mma(fa[i], fb, acc[kh * KS + kw][i][j]);
                }
            }
    }
    // D rows = co ((lane>>4)*4 + r), cols = ci (lane & 15)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TCI; ++j) {
                const int ci = bci + (wi * TCI + j) * 16 + i16;
                if (ci >= a.Cin) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = bco + (wc * TCO + i) * 16 + g * 4 + r;
                    if (co < a.Cout) atomicAdd(dwp + (long)co * a.Kpad + t * a.Cin + ci, acc[t][i][j][r]);
                }
            }
}

template <typename T, int KS, int S, int TCO, int TCI>
void launch(const WgArgs& a, const void* x, const void* dy, float* dwp, hipStream_t st) {
    const int cot = (a.Cout + 32 * TCO - 1) / (32 * TCO), cit = (a.Cin + 32 * TCI - 1) / (32 * TCI);
    WgArgs b = a;
    // each workgroup ends with KS*KS*(32*TCO)*(32*TCI) fp32 atomics (147 KB for a 3x3 64x64 tile): keep the
    // workgroup count at ~2 per CU for 3x3 so that flush stays far below the pixel traffic
    long want = (KS == 3 ? 512 : (TCO * TCI > 4 ? 1024 : 2048)) / ((long)cot * cit);
    // ... and give every workgroup at least ~16 patches (3x3) before it flushes: on small maps one-patch
    // workgroups spent 100+ us hammering the same 147 KB with atomics (64->64 3x3 @20x20: 117 us, 8 TFLOP/s)
    const long min_per = KS == 3 ? 16 : (TCO * TCI > 4 ? 8 : 4);
    if (want > a.npatch / min_per) want = a.npatch / min_per;
    if (want < 1) want = 1;
    b.per_slab = (a.npatch + want - 1) / want;
    const int nslab = (int)((a.npatch + b.per_slab - 1) / b.per_slab);
    hipLaunchKernelGGL((k_wgrad2<T, KS, S, TCO, TCI>), dim3(cot, cit, nslab), dim3(256), 0, st, b, (const T*)x, (const T*)dy, dwp);
}

template <typename T, int KS, int S>
void launch_tiles(const WgArgs& a, const void* x, const void* dy, float* dwp, hipStream_t st) {
    if constexpr (KS == 1) {
        // one tap: the accumulators of a (32*TCO x 32*TCI) tile are TCO*TCI*4 registers, so a workgroup can own
        // up to 128 x 128 of (co, ci) and X / dY are each read once for layers up to 128 channels
        const int to = a.Cout > 96 ? 4 : a.Cout > 64 ? 3 : a.Cout > 32 ? 2 : 1;
        const int ti = a.Cin > 96 ? 4 : a.Cin > 64 ? 3 : a.Cin > 32 ? 2 : 1;
#define WG_CASE(TO_, TI_) if (to == TO_ && ti == TI_) return launch<T, 1, 1, TO_, TI_>(a, x, dy, dwp, st)
        WG_CASE(1, 1); WG_CASE(1, 2); WG_CASE(1, 3); WG_CASE(1, 4);
        WG_CASE(2, 1); WG_CASE(2, 2); WG_CASE(2, 3); WG_CASE(2, 4);
        WG_CASE(3, 1); WG_CASE(3, 2); WG_CASE(3, 3); WG_CASE(3, 4);
        WG_CASE(4, 1); WG_CASE(4, 2); WG_CASE(4, 3); WG_CASE(4, 4);
#undef WG_CASE
    } else {
        const bool bigo = a.Cout > 32, bigi = a.Cin > 32;
        if (bigo && bigi) launch<T, KS, S, 2, 2>(a, x, dy, dwp, st);
        else if (bigo) launch<T, KS, S, 2, 1>(a, x, dy, dwp, st);
        else if (bigi) launch<T, KS, S, 1, 2>(a, x, dy, dwp, st);
        else launch<T, KS, S, 1, 1>(a, x, dy, dwp, st);
    }
}

template <typename T>
int launch_ks(const WgArgs& a, int k, int stride, const void* x, const void* dy, float* dwp, hipStream_t st) {
    if (k == 1 && stride == 1) launch_tiles<T, 1, 1>(a, x, dy, dwp, st);
    else if (k == 3 && stride == 1) launch_tiles<T, 3, 1>(a, x, dy, dwp, st);
    else if (k == 3 && stride == 2) launch_tiles<T, 3, 2>(a, x, dy, dwp, st);
    else return YOLO_ERR_ARG;
    return YOLO_LAUNCH_CHECK();
}

}  // namespace

// dwp (zeroed by the caller) += packed gradient; same eligibility as the first design (mfma_wgrad_eligible)
int mfma_wgrad2_launch(const void* x, int ldx, const void* dy, int ldy, float* dwp, int Kpad, int N, int H, int W, int Cin,
                       int OH, int OW, int Cout, int k, int stride, int dtype, hipStream_t st) {
    WgArgs a;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.OH = OH; a.OW = OW; a.Cout = Cout; a.ldy = ldy; a.Kpad = Kpad;
    a.pbh = (OH + 3) / 4; a.pbw = (OW + 7) / 8;
    a.npatch = (long)N * a.pbh * a.pbw;
    a.per_slab = 0;
    if (a.npatch == 0) return YOLO_OK;
    if (dtype == YOLO_BF16) return launch_ks<bf16_t>(a, k, stride, x, dy, dwp, st);
    if (dtype == YOLO_F16) return launch_ks<f16_t>(a, k, stride, x, dy, dwp, st);
    return YOLO_ERR_DTYPE;
}
