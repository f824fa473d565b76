// Device-side pieces shared by the conv kernels (conv_mfma.hip, conv_halo.hip): MFMA wrappers, the packed
// geometry passed by value, the XCD-aware tile order and the 16-lane row reduction.
#pragma once
#include <cstdlib>
#include "common.h"
#include "conv_geom.h"

int conv_wide_flag();            // conv_mfma.hip: 16-byte epilogue stores: 2 (default) exchange by v_permlane16_swap, 1 by ds_bpermute, 0 off (YOLO_CONV_WIDE, yolo_conv_wide_set)

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <typename T> struct mfma_ops;
template <> struct mfma_ops<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct mfma_ops<f16_t> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int BM = 128;    // destination pixels per workgroup
constexpr int BK = 32;     // K elements per step (one MFMA K)
constexpr int LDSROW = 32; // elements per LDS row: unpadded 64-byte rows whose four 16-byte chunks are XOR-swizzled
                           // by (-(row >> 2)) & 3 -- conflict-free for ds_read_b128 fragment reads (16 rows x 1 chunk
                           // per lane group) and for the ds_write_b128 staging (2 rows x 4 chunks per 8 lanes)

struct GeomDev {           // ConvGeom with the tap offsets packed (no dynamic indexing of kernargs)
    int N, Hs, Ws, Cs, lds, Hd, Wd, Cd, ldd, Hg, Wg, ostep, ooff_h, ooff_w, sstride, ntaps, KT, Kpad;
    unsigned dh_pack, dw_pack;   // 2 bits per tap: value + 1
    int tap_inner;               // MODE 2 K order: 1 = taps innermost, 0 = channel chunks innermost
    int dma;                     // MODE 2 -> 3: tiles go global -> LDS by LDS-DMA instead of through registers
    float* stats;                // optional [8][2][Cd] batch-statistics accumulator (forward of a BN conv)
    const void* acc2;            // ACC launches: second accumulate source (row stride ld2) or null
    int ld2;
    int wide;                    // 16-byte epilogue stores where the destination allows (YOLO_CONV_WIDE=0: the 8-byte form, A/B runs)
    int act;                     // inference epilogue: 1 = SiLU after the bias
    const void* res;             // inference epilogue: residual added after the activation (row stride ldr) or null
    int ldr;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// bijective XCD-aware remap (guide T1): blocks that share an XCD get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// LDS-DMA: 16 bytes per lane from a buffer descriptor straight into LDS at lds_base + 16*lane (wave-uniform base).
// Wrapped so that the host compilation pass never sees the device-only builtin.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, void* lds_base, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
#endif
}

// x + (x rotated by N lanes inside its row of 16): one VALU op (v_add_f32 with a DPP operand)
template <int N> __device__ __forceinline__ float row_ror_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float x) {
    x = row_ror_add<8>(x);
    x = row_ror_add<4>(x);
    x = row_ror_add<2>(x);
    return row_ror_add<1>(x);
}


// the inference epilogue on one group of four channels: v = act(v) (+ residual)
template <typename T>
__device__ __forceinline__ void fused_epilogue(float (&v)[4], int act, const void* res, long off) {
    if (act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * __frcp_rn(1.f + __expf(-v[r]));
    }
    if (res != nullptr) {
        float o[4];
        load_pack<T, 4>((const T*)res + off, o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += o[r];
    }
}

// value of lane ^ 16 (gfx950's v_permlane16_swap_b32 swaps the odd 16-lane rows of its first operand with the even rows of
// its second: with both operands = v, the first result holds v[lane - 16] in the odd rows, the second v[lane + 16] in the even
// rows; tests/test_gpu_selftest.py pins the mapping)
__device__ __forceinline__ unsigned swap_rows16(unsigned v, bool odd_row) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return odd_row ? r[0] : r[1];
}

// Epilogue of one output pixel: the WN 16-channel blocks of accumulator row `a` (lane: channels cq .. cq+3 of every block,
// pixel = the lane's fr) -> bias, inference act / residual, accumulate sources, store.  Shared by the five MFMA conv kernels.
// 16-byte stores where the destination allows it (base 16-byte aligned, row stride a multiple of 8 channels, G::wide): lanes
// (fr, fg) and (fr, fg ^ 1) hold neighbouring channel quads of the SAME pixel for every block j; one dword pair swapped
// between them (lane ^ 16) leaves the even lane with 8 consecutive channels of block j and the odd lane with 8 of block j+1 --
// half as many, twice as wide write requests per wave instruction (same-box A/B on the step with k_conv_mfma alone:
// 10.51 -> 10.38 ms; 512 -> 128 @80x80 data gradient 97 -> 78 us).  EVERY lane of the wave must call this (`live` false
// for pixels outside the map): the exchange is a cross-lane operation.
template <typename T, int WN, bool ACC, typename G>
__device__ __forceinline__ void store_pixel_blocks(const G& g, const f32x4 (&a)[WN], const float (&bv)[WN][4], T* __restrict__ dst,
                                                   long pix, bool live, int cbase, int cq, int lane) {
    T* drow = dst + pix * g.ldd;
    auto values = [&](int j, float (&v)[4]) {
        const int c = cbase + j * 16 + cq;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = a[j][r] + bv[j][r];
        if (!live || c >= g.Cd) return;
        if (g.act | (g.res != nullptr)) fused_epilogue<T>(v, g.act, g.res, pix * g.ldr + c);
        if (ACC) {
            float o[4];
            load_pack<T, 4>(drow + c, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
            if (g.acc2 != nullptr) {
                load_pack<T, 4>((const T*)g.acc2 + pix * g.ld2 + c, o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += o[r];
            }
        }
    };
    const bool wide = sizeof(T) == 2 && (WN % 2 == 0) && g.wide && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) && (g.ldd % 8 == 0);
    if (wide) {
        const bool odd = (lane >> 4) & 1;
#pragma unroll
        for (int j = 0; j + 1 < WN; j += 2) {
            float va[4], vb[4];
            values(j, va);
            values(j + 1, vb);
            pack_t<T, 4> pa, pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pa.v[r] = from_f<T>(va[r]); pb.v[r] = from_f<T>(vb[r]); }
            const uint2 ua = __builtin_bit_cast(uint2, pa), ub = __builtin_bit_cast(uint2, pb);
            const uint2 send = odd ? ua : ub;                      // what the partner keeps
            uint2 recv;
            if (g.wide == 2) {                                     // v_permlane16_swap: a VALU exchange of 16-lane rows, no LDS op
                recv.x = swap_rows16(send.x, odd);
                recv.y = swap_rows16(send.y, odd);
            } else {
                recv.x = __shfl_xor(send.x, 16, 64);
                recv.y = __shfl_xor(send.y, 16, 64);
            }
            // even lane: block j, channels cq .. cq+7 = own | partner's; odd lane: block j+1, channels cq-4 .. cq+3
            const uint4 out = odd ? make_uint4(recv.x, recv.y, ub.x, ub.y) : make_uint4(ua.x, ua.y, recv.x, recv.y);
            const int c8 = cbase + (odd ? j + 1 : j) * 16 + (odd ? cq - 4 : cq);
            if (live && c8 < g.Cd) *reinterpret_cast<uint4*>(drow + c8) = out;
        }
    } else if (live) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int c = cbase + j * 16 + cq;
            if (c >= g.Cd) continue;            // Cd % 8 == 0 => a group of 4 is all-in or all-out
            float v[4];
            values(j, v);
            store_pack<T, 4>(drow + c, v);
        }
    }
}

inline GeomDev to_dev(const ConvGeom& g) {
    GeomDev d;
    d.N = g.N; d.Hs = g.Hs; d.Ws = g.Ws; d.Cs = g.Cs; d.lds = g.lds; d.Hd = g.Hd; d.Wd = g.Wd; d.Cd = g.Cd;
    d.ldd = g.ldd; d.Hg = g.Hg; d.Wg = g.Wg; d.ostep = g.ostep; d.ooff_h = g.ooff_h; d.ooff_w = g.ooff_w;
    d.sstride = g.sstride; d.ntaps = g.ntaps; d.Kpad = g.Kpad; d.KT = g.Kpad / BK;
    d.dh_pack = d.dw_pack = 0;
    d.stats = g.stats;
    d.acc2 = g.acc2; d.ld2 = g.ld2;
    d.act = g.act; d.res = g.res; d.ldr = g.ldr;
    d.wide = conv_wide_flag();
    d.tap_inner = 0;
    d.dma = 1;      // LDS-DMA staging: level or a few % ahead of register staging on every shape of tools/conv_tune.py
    for (int t = 0; t < g.ntaps; ++t) {
        d.dh_pack |= (unsigned)(g.dh[t] + 1) << (2 * t);
        d.dw_pack |= (unsigned)(g.dw[t] + 1) << (2 * t);
    }
    return d;
}


}  // namespace
