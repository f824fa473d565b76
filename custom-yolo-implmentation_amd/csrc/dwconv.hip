// Depthwise 3x3 (stride 1, pad 1) for 16-bit NHWC tensors: forward, data gradient (flipped taps) and weight gradient.
//
// The first form (k_dw3x3 in conv_generic.hip, kept for fp32 and unaligned tensors) fetched nine 16-byte neighbours
// per output through divergent border branches: 60-100 us on the 80x80x128 maps of the head against 21 us of HBM time.
// Here a thread owns 8 channels (one 16-byte packet) of a vertical strip of R output pixels: (R+2) x 3 packets are
// fetched up front through a buffer descriptor -- border packets are offsets beyond the descriptor's range, which the
// hardware returns as zeros, so there is no branch between the loads and all of them are in flight together -- and
// feed R x 9 x 8 FMAs: 4.5 loads per output for R = 4 instead of 9, horizontal reuse through the vector L1 (lanes of
// a wave that differ in the column fetch overlapping 1-KB runs).  The 9 x C fp32 taps of the workgroup's channels
// sit in LDS ([tap][half][channel group][4]: conflict-free 16-byte reads).
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ void unpack8(const u32x4& v, float (&o)[8]);
template <> __device__ __forceinline__ void unpack8<bf16_t>(const u32x4& v, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(v[i] << 16);
        o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack8<f16_t>(const u32x4& v, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (a bit_cast of the vector ELEMENT to a half2 was miscompiled for i >= 1: channels 2..7 of every packet came out
        // wrong in fp16 while bf16 was exact -- found by the FSDP2 fp16 parity test, tests/test_gpu_kernels.py::test_depthwise)
        const unsigned int u = v[i];
        o[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
        o[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
    }
}

struct DwGeom {
    int N, H, W, C, ldx, ldy;
    int cvb;        // channel groups (of 8) per workgroup
    int ncol;       // columns per workgroup
    int nsh;        // row strips per image
    int ncb;        // column blocks per row
    long total;     // strips * column blocks * images
    const float* bias;   // fused inference (forward only): y = act(conv + bias); nullptr / 0 otherwise
    int act;
};

constexpr int OOB = (int)0x80000000;

// taps of the workgroup's channels -> LDS, [tap][half][cvb][4]
template <bool FLIP>
__device__ __forceinline__ void stage_taps(float* wl, const float* __restrict__ w, int c0, int C, int cvb) {
    for (int i = threadIdx.x; i < cvb * 8 * 9; i += 256) {
        const int ch = i / 9, t = i - ch * 9;
        const float v = (c0 + ch < C) ? w[(long)(c0 + ch) * 9 + (FLIP ? 8 - t : t)] : 0.f;
        wl[((t * 2 + ((ch >> 2) & 1)) * cvb + (ch >> 3)) * 4 + (ch & 3)] = v;
    }
}

template <typename T, int R, bool FLIP>
__global__ __launch_bounds__(256) void k_dw3x3_strip(DwGeom g, const T* __restrict__ x, const float* __restrict__ w,
                                                     T* __restrict__ y, int accumulate, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int c0 = blockIdx.y * g.cvb * 8;
    stage_taps<FLIP>(wl, w, c0, g.C, g.cvb);
    __syncthreads();
    const int cgl = threadIdx.x % g.cvb, col = threadIdx.x / g.cvb;
    const int ch = c0 + cgl * 8;
    const bool active = col < g.ncol && ch < g.C;
    float ssum[8], ssq[8];                                   // BatchNorm batch statistics of the stored (rounded) values
#pragma unroll
    for (int j = 0; j < 8; ++j) ssum[j] = ssq[j] = 0.f;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, g.N * g.H * g.W * g.ldx * 2, 0x00020000);
    const float4* wq = reinterpret_cast<const float4*>(wl) + cgl;
    for (long s = blockIdx.x; active && s < g.total; s += gridDim.x) {
        const int cbk = (int)(s % g.ncb);
        const long t2 = s / g.ncb;
        const int sh = (int)(t2 % g.nsh), n = (int)(t2 / g.nsh);
        const int h0 = sh * R, wc = cbk * g.ncol + col;
        if (wc >= g.W) continue;
        u32x4 v[R + 2][3];
#pragma unroll
        for (int r = 0; r < R + 2; ++r) {
            const int hs = h0 - 1 + r;
            const bool rok = (unsigned)hs < (unsigned)g.H;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ws = wc - 1 + c;
                const bool ok = rok && (unsigned)ws < (unsigned)g.W;
                const int off = ok ? ((((n * g.H + hs) * g.W + ws) * g.ldx + ch) * 2) : OOB;
                v[r][c] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            }
        }
        float acc[R][8];
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
#pragma unroll
        for (int r = 0; r < R + 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float f[8];
                unpack8<T>(v[r][c], f);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int o = r - kh;
                    if (o < 0 || o >= R) continue;
                    const float4 w0 = wq[((kh * 3 + c) * 2) * g.cvb], w1 = wq[((kh * 3 + c) * 2 + 1) * g.cvb];
                    acc[o][0] = fmaf(f[0], w0.x, acc[o][0]); acc[o][1] = fmaf(f[1], w0.y, acc[o][1]);
                    acc[o][2] = fmaf(f[2], w0.z, acc[o][2]); acc[o][3] = fmaf(f[3], w0.w, acc[o][3]);
                    acc[o][4] = fmaf(f[4], w1.x, acc[o][4]); acc[o][5] = fmaf(f[5], w1.y, acc[o][5]);
                    acc[o][6] = fmaf(f[6], w1.z, acc[o][6]); acc[o][7] = fmaf(f[7], w1.w, acc[o][7]);
                }
            }
#pragma unroll
        for (int o = 0; o < R; ++o) {
            if (h0 + o >= g.H) break;
            T* dst = y + ((long)(n * g.H + h0 + o) * g.W + wc) * g.ldy + ch;
            if (!FLIP && g.bias != nullptr) {                 // folded BatchNorm (Model.fuse()): bias, then SiLU
                const float4 b0 = *reinterpret_cast<const float4*>(g.bias + ch), b1 = *reinterpret_cast<const float4*>(g.bias + ch + 4);
                acc[o][0] += b0.x; acc[o][1] += b0.y; acc[o][2] += b0.z; acc[o][3] += b0.w;
                acc[o][4] += b1.x; acc[o][5] += b1.y; acc[o][6] += b1.z; acc[o][7] += b1.w;
                if (g.act) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] = acc[o][j] * __frcp_rn(1.f + __expf(-acc[o][j]));
                }
            }
            if (accumulate) {                                 // gradient fan-in: add to what another consumer's backward left
                float prev[8];
                load_pack<T, 8>(dst, prev);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] += prev[j];
            }
            store_pack<T, 8>(dst, acc[o]);
            if (!FLIP && stats != nullptr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float r2 = to_f<T>(from_f<T>(acc[o][j]));
                    ssum[j] += r2;
                    ssq[j] += r2 * r2;
                }
            }
        }
    }
    if (FLIP || stats == nullptr) return;
    // workgroup totals per channel (the columns of one channel group through LDS), then one atomic per channel and
    // statistic into replica (workgroup index mod 8) of the [8][2][C] accumulator -- the forward of a BatchNorm conv
    // needs no separate statistics pass over y
    __shared__ float red[256][9];
    float* o = stats + (long)(blockIdx.x & 7) * 2 * g.C;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = active ? (pass ? ssq[j] : ssum[j]) : 0.f;
        __syncthreads();
        for (int e = threadIdx.x; e < g.cvb * 8; e += 256) {
            const int cl = e >> 3, j = e & 7;
            float t = 0.f;
            for (int cc = 0; cc < g.ncol; ++cc) t += red[cc * g.cvb + cl][j];
            if (c0 + e < g.C) atomicAdd(o + pass * g.C + c0 + e, t);
        }
    }
}

// dw[c][tap] = sum_p dy[p][c] * x[p (+) tap][c].  Same strips; a thread keeps the 9 x 8 sums of its channels over all the
// strips its workgroup walks, the workgroup combines its columns through LDS one tap at a time and stores ITS partial
// [C][9] block (plain stores; k_dw_wgrad_finalize sums the workgroups).
template <typename T, int R>
__global__ __launch_bounds__(256) void k_dw3x3_wgrad_strip(DwGeom g, const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* __restrict__ part) {
    __shared__ float red[256][9];
    const int c0 = blockIdx.y * g.cvb * 8;
    const int cgl = threadIdx.x % g.cvb, col = threadIdx.x / g.cvb;
    const int ch = c0 + cgl * 8;
    const bool active = col < g.ncol && ch < g.C;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    const __amdgpu_buffer_rsrc_t rsx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, g.N * g.H * g.W * g.ldx * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy), 0, g.N * g.H * g.W * g.ldy * 2, 0x00020000);
    if (active) {
        for (long s = blockIdx.x; s < g.total; s += gridDim.x) {
            const int cbk = (int)(s % g.ncb);
            const long t2 = s / g.ncb;
            const int sh = (int)(t2 % g.nsh), n = (int)(t2 / g.nsh);
            const int h0 = sh * R, wc = cbk * g.ncol + col;
            if (wc >= g.W) continue;
            u32x4 gv[R], v[R + 2][3];
#pragma unroll
            for (int o = 0; o < R; ++o) {
                const bool ok = h0 + o < g.H;
                gv[o] = __builtin_amdgcn_raw_buffer_load_b128(rsy, ok ? ((((n * g.H + h0 + o) * g.W + wc) * g.ldy + ch) * 2) : OOB, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < R + 2; ++r) {
                const int hs = h0 - 1 + r;
                const bool rok = (unsigned)hs < (unsigned)g.H;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ws = wc - 1 + c;
                    const bool ok = rok && (unsigned)ws < (unsigned)g.W;
                    v[r][c] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? ((((n * g.H + hs) * g.W + ws) * g.ldx + ch) * 2) : OOB, 0, 0);
                }
            }
            float gf[R][8];
#pragma unroll
            for (int o = 0; o < R; ++o) unpack8<T>(gv[o], gf[o]);
#pragma unroll
            for (int r = 0; r < R + 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float f[8];
                    unpack8<T>(v[r][c], f);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int o = r - kh;
                        if (o < 0 || o >= R) continue;
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[kh * 3 + c][j] = fmaf(gf[o][j], f[j], acc[kh * 3 + c][j]);
                    }
                }
        }
    }
    float* pw = part + (long)blockIdx.x * g.C * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = active ? acc[t][j] : 0.f;
        __syncthreads();
        for (int o = threadIdx.x; o < g.cvb * 8; o += 256) {
            const int cl = o >> 3, j = o & 7;
            float s = 0.f;
            for (int cc = 0; cc < g.ncol; ++cc) s += red[cc * g.cvb + cl][j];
            if (c0 + o < g.C) pw[(long)(c0 + o) * 9 + t] = s;
        }
    }
}

bool dw_geom(DwGeom& g, const void* x, int ldx, const void* y, int ldy, int N, int H, int W, int C, int R) {
    if (C % 8 || ldx % 8 || ldy % 8 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return false;
    if ((long)N * H * W * ldx * 2 >= (1L << 31) || (long)N * H * W * ldy * 2 >= (1L << 31)) return false;
    g.N = N; g.H = H; g.W = W; g.C = C; g.ldx = ldx; g.ldy = ldy;
    const int cv = C / 8;
    g.cvb = cv < 256 ? cv : 256;
    g.ncol = 256 / g.cvb;
    if (g.ncol > W) g.ncol = W;
    g.nsh = (H + R - 1) / R;
    g.ncb = (W + g.ncol - 1) / g.ncol;
    g.total = (long)N * g.nsh * g.ncb;
    g.bias = nullptr; g.act = 0;
    return g.total > 0;
}

}  // namespace

// forward / data gradient; returns -1 when the tensors do not qualify (the caller falls back to the generic kernel)
int dw_strip_launch(bool flip, const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C, int dtype,
                    int accumulate, float* stats, hipStream_t st, const float* bias, int act) {
    if (dtype != YOLO_BF16 && dtype != YOLO_F16) return -1;
    constexpr int R = 4;
    DwGeom g;
    if (!dw_geom(g, x, ldx, y, ldy, N, H, W, C, R)) return -1;
    if (bias != nullptr && (flip || stats != nullptr || (reinterpret_cast<uintptr_t>(bias) & 15))) return -1;
    g.bias = bias; g.act = act;
    const dim3 grid((unsigned)(g.total < 4096 ? g.total : 4096), (unsigned)ceil_div(C / 8, g.cvb));
    const size_t lds = (size_t)g.cvb * 8 * 9 * sizeof(float);
#define DW_GO(T_, FLIP_) hipLaunchKernelGGL((k_dw3x3_strip<T_, R, FLIP_>), grid, dim3(256), lds, st, g, (const T_*)x, w, (T_*)y, accumulate, stats)
    if (dtype == YOLO_BF16) { if (flip) DW_GO(bf16_t, true); else DW_GO(bf16_t, false); }
    else { if (flip) DW_GO(f16_t, true); else DW_GO(f16_t, false); }
#undef DW_GO
    return YOLO_LAUNCH_CHECK();
}

// partial: [nslab][C][9] fp32, every entry written; -1 when the tensors do not qualify
int dw_strip_wgrad_launch(const void* x, int ldx, const void* dy, int ldy, float* partial, int nslab, int N, int H, int W, int C,
                          int dtype, hipStream_t st) {
    if (dtype != YOLO_BF16 && dtype != YOLO_F16) return -1;
    constexpr int R = 2;
    DwGeom g;
    if (!dw_geom(g, x, ldx, dy, ldy, N, H, W, C, R)) return -1;
    const dim3 grid((unsigned)nslab, (unsigned)ceil_div(C / 8, g.cvb));
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((k_dw3x3_wgrad_strip<bf16_t, R>), grid, dim3(256), 0, st, g, (const bf16_t*)x, (const bf16_t*)dy, partial);
    else
        hipLaunchKernelGGL((k_dw3x3_wgrad_strip<f16_t, R>), grid, dim3(256), 0, st, g, (const f16_t*)x, (const f16_t*)dy, partial);
    return YOLO_LAUNCH_CHECK();
}
