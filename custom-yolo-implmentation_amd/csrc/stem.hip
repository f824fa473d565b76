// Stem (first layer: 3 -> C, 3x3 stride 2 pad 1, reference src/model/backbone.py:38).
// Cin = 3 cannot feed the MFMA gather (8-channel packets), and a VALU direct conv cost 8 ms per step.
// Instead the image is unfolded ONCE into a K = 27 (+5 zero) column matrix -- reading the NCHW input
// tensor directly, so the NCHW->NHWC conversion disappears too -- and forward / weight-gradient run
// as 1x1 convolutions on the MFMA kernels.  k = ci*9 + kh*3 + kw is exactly the OIHW flattening, so
// weights and weight gradients are plain [Cout][27] views padded to 32.
#include "common.h"
#include "conv_dev.h"

namespace {

template <typename TI, typename TO>
__global__ void k_stem_im2col(const TI* __restrict__ img, TO* __restrict__ col, int N, int H, int W, int OH, int OW) {
    long total = (long)N * OH * OW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        int ow = (int)(p % OW);
        long t = p / OW;
        int oh = (int)(t % OH);
        long n = t / OH;
        float v[32];
#pragma unroll
        for (int k = 27; k < 32; ++k) v[k] = 0.f;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const TI* plane = img + (n * 3 + ci) * (long)H * W;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                int ih = 2 * oh + kh - 1;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    int iw = 2 * ow + kw - 1;
                    bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
                    v[ci * 9 + kh * 3 + kw] = ok ? to_f<TI>(plane[(long)ih * W + iw]) : 0.f;
                }
            }
        }
        TO* o = col + p * 32;
        constexpr int V = vec_of<TO>::N;
#pragma unroll
        for (int k = 0; k < 32; k += V) {
            float w[V];
#pragma unroll
            for (int j = 0; j < V; ++j) w[j] = v[k + j];
            store_pack<TO, V>(o + k, w);
        }
    }
}

template <typename P, typename T>
__global__ void k_stem_pack_w(const P* __restrict__ w, int Cout, T* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cout * 32) return;
    int o = e >> 5, k = e & 31;
    out[e] = from_f<T>(k < 27 ? to_f<P>(w[o * 27 + k]) : 0.f);
}

template <typename P>
__global__ void k_stem_unpack_dw(const float* __restrict__ dw32, int Cout, P* __restrict__ dw) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cout * 27) return;
    int o = e / 27, k = e - o * 27;
    dw[e] = from_f<P>(dw32[o * 32 + k]);
}

// ---- fused forward: y = W * unfold(img) straight from the NCHW fp32 image (no column tensor in HBM) --------------------
// A workgroup owns 128 consecutive output pixels of TWO output rows: the 3 channels x 5 input rows it needs (257
// columns each) are staged once in LDS, one 16-byte load per lane = one row segment per wave instruction; every lane then gathers its MFMA B fragment
// (8 consecutive k of one pixel; k = ci*9 + kh*3 + kw, so row = k/3 and column offset = k%3) from LDS and the A
// fragments (weights [Cout][32], L2-resident) from global memory: one 16x16x32 MFMA per (16 channels x 16 pixels).
// Epilogue as k_conv_mfma: 8-byte channel groups per lane, BatchNorm batch statistics of the rounded values.
constexpr int STEM_SEG = 128;
constexpr int STEM_ROWW = 2 * STEM_SEG + 1;
constexpr int STEM_IR = 5;          // input rows per channel for TWO output rows (the middle one is shared)

template <typename T, int CT>
__global__ __launch_bounds__(256) void k_stem_conv(const float* __restrict__ img, const T* __restrict__ wp,
                                                   T* __restrict__ y, int ldy, float* __restrict__ stats, int N, int H,
                                                   int W, int OH, int OW, int Cout, const float* __restrict__ bias, int act) {
    typedef mfma_ops<T> ops;
    typedef typename ops::frag frag;
    // rows[ci*5 + ir][4 + c] = input row 2*oh0 - 1 + ir, column 2*ow0 + c (c in [0, 256)); [..][3] = column 2*ow0 - 1
    __shared__ __attribute__((aligned(16))) float rows[3 * STEM_IR][STEM_ROWW + 7];
    __shared__ float sacc[2][16 * CT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kg = lane >> 4;
    const int segs = (OW + STEM_SEG - 1) / STEM_SEG, ohp = (OH + 1) >> 1;
    const int seg = blockIdx.x % segs;
    const int t = blockIdx.x / segs;
    const int oh0 = (t % ohp) * 2, n = t / ohp;
    const int ow0 = seg * STEM_SEG, iw0 = 2 * ow0 - 1;
    if ((W & 3) == 0) {
        // one 16-byte load per lane = one whole row segment per wave instruction
        // a wave's (up to four) row segments: all loads in flight before the first LDS store (clamped addresses, no branch)
        constexpr int RPW = (3 * STEM_IR + 3) / 4;
        float4 v[RPW];
        float edge[RPW];
        const int c = lane * 4;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int r = wave + 4 * k, rr = r < 3 * STEM_IR ? r : 0;
            const int ci = rr / STEM_IR, ir = rr - ci * STEM_IR;
            const int ih = 2 * oh0 + ir - 1;
            const bool rok = ih >= 0 && ih < H;
            const float* src = img + (((long)n * 3 + ci) * H + (rok ? ih : 0)) * (long)W + 2 * ow0;
            const bool cok = 2 * ow0 + c < W;                                               // W % 4 == 0: all four or none
            v[k] = *reinterpret_cast<const float4*>(src + (cok ? c : 0));
            if (!(rok && cok)) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            edge[k] = (lane == 0 && ow0 > 0) ? src[-1] : 0.f;
            if (!rok) edge[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int r = wave + 4 * k;
            if (r < 3 * STEM_IR) {
                *reinterpret_cast<float4*>(&rows[r][4 + c]) = v[k];
                if (lane == 0) rows[r][3] = edge[k];
            }
        }
    } else {
        for (int idx = tid; idx < 3 * STEM_IR * STEM_ROWW; idx += 256) {
            const int r = idx / STEM_ROWW, c = idx - r * STEM_ROWW;
            const int ci = r / STEM_IR, ir = r - ci * STEM_IR;
            const int ih = 2 * oh0 + ir - 1, iw = iw0 + c;
            const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
            rows[r][3 + c] = ok ? img[(((long)n * 3 + ci) * H + ih) * (long)W + iw] : 0.f;
        }
    }
    for (int c = tid; c < 2 * 16 * CT; c += 256) (&sacc[0][0])[c] = 0.f;
    frag a[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) a[ct] = *reinterpret_cast<const frag*>(wp + (long)(ct * 16 + fr) * 32 + kg * 8);
    __syncthreads();
    // a wave owns four 16-pixel tiles: tile id wave*4 + i = (output row j) * 8 + (tile within the 128-pixel segment)
    f32x4 acc[CT][4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tl = wave * 4 + i, j = tl >> 3;
        const int p = (tl & 7) * 16 + fr;                   // pixel of this lane's B column
        live[i] = ow0 + p < OW && oh0 + j < OH;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kg * 8 + e;
            const int ci = k / 9, r3 = k / 3, kh = r3 - ci * 3, kw = k - r3 * 3;
            v[e] = (k < 27 && live[i]) ? rows[k < 27 ? ci * STEM_IR + 2 * j + kh : 0][2 * p + kw + 3] : 0.f;
        }
        frag b;
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = from_f<T>(v[e]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            acc[ct][i] = ops::mma(a[ct], b, z);
        }
    }
    // lane holds channels ct*16 + kg*4 .. +3 of its tile's pixel fr
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!live[i]) continue;
        const int tl = wave * 4 + i;
        T* drow = y + (((long)n * OH + oh0 + (tl >> 3)) * OW + ow0 + (tl & 7) * 16 + fr) * (long)ldy;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ct][i][r];
            if (bias != nullptr) {                              // fused inference (BatchNorm folded in): act(conv + b)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bias[ct * 16 + kg * 4 + r];
            }
            if (act) fused_epilogue<T>(v, act, nullptr, 0);
            store_pack<T, 4>(drow + ct * 16 + kg * 4, v);
        }
    }
    if (stats != nullptr) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float s[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = to_f<T>(from_f<T>(acc[ct][i][r]));      // pixels past the row / image end hold 0
                    s[r] += v;
                    q2[r] += v * v;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r] = row16_sum(s[r]);
                q2[r] = row16_sum(q2[r]);
            }
            if (fr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atomicAdd(&sacc[0][ct * 16 + kg * 4 + r], s[r]);
                    atomicAdd(&sacc[1][ct * 16 + kg * 4 + r], q2[r]);
                }
            }
        }
        __syncthreads();
        float* o = stats + (long)(blockIdx.x & 7) * 2 * Cout;
        for (int c = tid; c < 16 * CT; c += 256) {
            atomicAdd(o + c, sacc[0][c]);
            atomicAdd(o + Cout + c, sacc[1][c]);
        }
    }
}

// ---- weight gradient straight from the NCHW image: dW[co][k] = sum_p dY[p][co] * unfold(img)[p][k] -----------------
// Same tile as the forward (two output rows x 128 pixels, the 3 x 5 input row segments staged once in LDS) plus the
// tile's dY rows in their natural [pixel][channel] order.  The reduction index is the pixel: per group of 32 consecutive
// output pixels the dY^T fragments come out of LDS through the transposing read (as in k_wgrad2) and the unfold
// fragments are gathered from the staged image rows (8 consecutive pixels of one k = (ci, kh, kw) per lane, rounded
// to the compute dtype exactly as the column tensor was).  A workgroup walks tiles with a grid stride, keeps
// [Cout][32] sums per wave in registers, combines its waves in LDS and stores ONE partial matrix;
// k_stem_wgrad_reduce sums the workgroups straight into the OIHW gradient.  No column tensor exists any more.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <typename T>
__device__ __forceinline__ typename mfma_ops<T>::frag stem_tr_frag(const T* p_lo, const T* p_hi) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_lo);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_hi);
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename mfma_ops<T>::frag, both);
}

constexpr int stem_ldy(int ct) { return ct == 4 ? 72 : ct == 8 ? 144 : ct * 16 + (ct == 2 ? 0 : 8); }

template <typename T, int CT>
__global__ __launch_bounds__(256) void k_stem_wgrad(const float* __restrict__ img, const T* __restrict__ dy, int ldy,
                                                    float* __restrict__ part, int N, int H, int W, int OH, int OW, long tiles) {
    typedef mfma_ops<T> ops;
    typedef typename ops::frag frag;
    constexpr int LDY = stem_ldy(CT), COUT = 16 * CT;
    __shared__ __attribute__((aligned(16))) float rows[3 * STEM_IR][STEM_ROWW + 7];
    __shared__ __attribute__((aligned(16))) T ys[2 * STEM_SEG * LDY];
    __shared__ float sum[COUT * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, c4 = 4 * (i16 & 3);
    const int segs = (OW + STEM_SEG - 1) / STEM_SEG, ohp = (OH + 1) >> 1;
    f32x4 acc[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[ct][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's two unfold columns: k = nt*16 + i16 -> (ci, kh, kw); k >= 27 reads nothing
    int krow[2], kcol[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int k = nt * 16 + i16, ci = k / 9, r3 = k / 3;
        krow[nt] = k < 27 ? ci * STEM_IR + (r3 - ci * 3) : -1;
        kcol[nt] = k - r3 * 3 + 3;
    }
    for (long tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const int seg = (int)(tl % segs);
        const long t = tl / segs;
        const int oh0 = (int)(t % ohp) * 2, n = (int)(t / ohp);
        const int ow0 = seg * STEM_SEG, iw0 = 2 * ow0 - 1;
        __syncthreads();                                    // the previous tile's fragments have been read
        if ((W & 3) == 0) {
            constexpr int RPW = (3 * STEM_IR + 3) / 4;            // a wave's row segments: all loads before the first LDS store
            float4 v[RPW];
            float edge[RPW];
            const int c = lane * 4;
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int r = wave + 4 * k, rr = r < 3 * STEM_IR ? r : 0;
                const int ci = rr / STEM_IR, ir = rr - ci * STEM_IR;
                const int ih = 2 * oh0 + ir - 1;
                const bool rok = ih >= 0 && ih < H;
                const float* src = img + (((long)n * 3 + ci) * H + (rok ? ih : 0)) * (long)W + 2 * ow0;
                const bool cok = 2 * ow0 + c < W;
                v[k] = *reinterpret_cast<const float4*>(src + (cok ? c : 0));
                if (!(rok && cok)) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                edge[k] = (lane == 0 && ow0 > 0) ? src[-1] : 0.f;
                if (!rok) edge[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int r = wave + 4 * k;
                if (r < 3 * STEM_IR) {
                    *reinterpret_cast<float4*>(&rows[r][4 + c]) = v[k];
                    if (lane == 0) rows[r][3] = edge[k];
                }
            }
        } else {
            for (int idx = tid; idx < 3 * STEM_IR * STEM_ROWW; idx += 256) {
                const int r = idx / STEM_ROWW, c = idx - r * STEM_ROWW;
                const int ci = r / STEM_IR, ir = r - ci * STEM_IR;
                const int ih = 2 * oh0 + ir - 1, iw = iw0 + c;
                const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
                rows[r][3 + c] = ok ? img[(((long)n * 3 + ci) * H + ih) * (long)W + iw] : 0.f;
            }
        }
        // dY: 2 rows x 128 pixels x COUT channels, 16-byte chunks; pixels past the row / image end are zeros
        constexpr int CPP = COUT / 8;
        constexpr int DYR = 2 * STEM_SEG * CPP / 256;         // chunks per thread (CPP = COUT / 8: 2 ... 16), all in flight
        uint4 dv[DYR];
#pragma unroll
        for (int k = 0; k < DYR; ++k) {
            const int id = tid + 256 * k;
            const int px = id / CPP, ch = (id - px * CPP) * 8;
            const int j = px / STEM_SEG, p = px - j * STEM_SEG;
            const bool ok = oh0 + j < OH && ow0 + p < OW;
            dv[k] = *reinterpret_cast<const uint4*>(dy + (((long)n * OH + (ok ? oh0 + j : 0)) * OW + (ok ? ow0 + p : 0)) * (long)ldy + ch);
            if (!ok) dv[k] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < DYR; ++k) {
            const int id = tid + 256 * k;
            const int px = id / CPP, ch = (id - px * CPP) * 8;
            *reinterpret_cast<uint4*>(ys + px * LDY + ch) = dv[k];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {                       // wave w: the 32-pixel group w of both output rows
            const int px0 = wave * 32;
            frag fb[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = krow[nt] >= 0 ? rows[krow[nt] < 0 ? 0 : krow[nt] + 2 * j][2 * (px0 + 8 * g + e) + kcol[nt]] : 0.f;
                    fb[nt][e] = from_f<T>(v);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const T* p = ys + (j * STEM_SEG + px0 + 8 * g + q) * LDY + ct * 16 + c4;
                const frag fa = stem_tr_frag<T>(p, p + 4 * LDY);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[ct][nt] = ops::mma(fa, fb[nt], acc[ct][nt]);
            }
        }
    }
    // D rows = co (ct*16 + g*4 + r), cols = k (nt*16 + i16)
    // the four waves add their sums one after the other (fixed order: the result does not depend on the schedule)
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* e = &sum[(ct * 16 + g * 4 + r) * 32 + nt * 16 + i16];
                        *e = wv == 0 ? acc[ct][nt][r] : *e + acc[ct][nt][r];
                    }
        }
    }
    __syncthreads();
    float* o = part + (long)blockIdx.x * COUT * 32;
    for (int c = tid; c < COUT * 32; c += 256) o[c] = sum[c];
}

// dw[o][k] (OIHW (Cout,3,3,3) flat, k < 27) = sum over the workgroups' partial matrices; 32 lanes per element
template <typename P>
__global__ __launch_bounds__(256) void k_stem_wgrad_reduce(const float* __restrict__ part, int nslab, int Cout, P* __restrict__ dw) {
    const int e = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (e >= Cout * 27) return;
    const int o = e / 27, k = e - o * 27;
    float s = 0.f;
    for (int b = l; b < nslab; b += 32) s += part[((long)b * Cout + o) * 32 + k];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 32);
    if (l == 0) dw[e] = from_f<P>(s);
}

constexpr int STEM_WG_SLABS = 1024;

template <typename T>
int launch_stem_wgrad(const float* img, const void* dy, int ldy, float* part, int N, int H, int W, int OH, int OW, int Cout,
                      int* nslab, hipStream_t st) {
    const long tiles = (long)N * ((OH + 1) / 2) * ((OW + STEM_SEG - 1) / STEM_SEG);
    if (tiles <= 0) return YOLO_ERR_ARG;
    const int grid = (int)(tiles < STEM_WG_SLABS ? tiles : STEM_WG_SLABS);
    *nslab = grid;
#define STEM_WG(CT) hipLaunchKernelGGL((k_stem_wgrad<T, CT>), dim3(grid), dim3(256), 0, st, img, (const T*)dy, ldy, part, N, H, W, OH, OW, tiles)
    switch (Cout / 16) {
        case 1: STEM_WG(1); break;
        case 2: STEM_WG(2); break;
        case 3: STEM_WG(3); break;
        case 4: STEM_WG(4); break;
        case 6: STEM_WG(6); break;
        case 8: STEM_WG(8); break;
        default: return YOLO_ERR_ARG;
    }
#undef STEM_WG
    return YOLO_LAUNCH_CHECK();
}

template <typename T>
int launch_stem_conv(const float* img, const void* wp, void* y, int ldy, float* stats, int N, int H, int W, int OH, int OW,
                     int Cout, const float* bias, int act, hipStream_t st) {
    const long blocks = (long)N * ((OH + 1) / 2) * ((OW + STEM_SEG - 1) / STEM_SEG);
    if (blocks <= 0 || blocks > 0x7fffffffL) return YOLO_ERR_ARG;
#define STEM_CT(CT) hipLaunchKernelGGL((k_stem_conv<T, CT>), dim3((unsigned)blocks), dim3(256), 0, st, img, (const T*)wp, (T*)y, ldy, stats, N, H, W, OH, OW, Cout, bias, act)
    switch (Cout / 16) {
        case 1: STEM_CT(1); break;
        case 2: STEM_CT(2); break;
        case 3: STEM_CT(3); break;
        case 4: STEM_CT(4); break;
        case 6: STEM_CT(6); break;
        case 8: STEM_CT(8); break;
        default: return YOLO_ERR_ARG;
    }
#undef STEM_CT
    return YOLO_LAUNCH_CHECK();
}

}  // namespace

extern "C" {

// img: (N,3,H,W) contiguous NCHW of img_dtype; col: [N*OH*OW][32] of col_dtype (an NHWC tensor with C = 32)
int yolo_stem_im2col(const void* img, int img_dtype, void* col, int col_dtype, int N, int H, int W, int OH, int OW,
                     hipStream_t st) {
    if (OH != (H - 1) / 2 + 1 || OW != (W - 1) / 2 + 1) return YOLO_ERR_ARG;
    long total = (long)N * OH * OW;
    int grid = (int)((total + 255) / 256 > 256 * 16 ? 256 * 16 : (total + 255) / 256);
    if (grid < 1) grid = 1;
#define STEM_LAUNCH(TI) YOLO_DISPATCH_T(col_dtype, hipLaunchKernelGGL((k_stem_im2col<TI, T>), dim3(grid), dim3(256), 0, st, (const TI*)img, (T*)col, N, H, W, OH, OW))
    switch (img_dtype) {
        case YOLO_F32:  STEM_LAUNCH(float); break;
        case YOLO_BF16: STEM_LAUNCH(bf16_t); break;
        case YOLO_F16:  STEM_LAUNCH(f16_t); break;
        default: return YOLO_ERR_DTYPE;
    }
#undef STEM_LAUNCH
    return YOLO_LAUNCH_CHECK();
}

// w: OIHW (Cout,3,3,3) -> out [Cout][32] (the forward-packed matrix of a 1x1 conv with Cin = 32)
int yolo_stem_pack_weights(const void* w, int w_dtype, int Cout, void* out, int out_dtype, hipStream_t st) {
    int grid = (Cout * 32 + 255) / 256;
#define PACK_LAUNCH(P) YOLO_DISPATCH_T(out_dtype, hipLaunchKernelGGL((k_stem_pack_w<P, T>), dim3(grid), dim3(256), 0, st, (const P*)w, Cout, (T*)out))
    switch (w_dtype) {
        case YOLO_F32:  PACK_LAUNCH(float); break;
        case YOLO_BF16: PACK_LAUNCH(bf16_t); break;
        case YOLO_F16:  PACK_LAUNCH(f16_t); break;
        default: return YOLO_ERR_DTYPE;
    }
#undef PACK_LAUNCH
    return YOLO_LAUNCH_CHECK();
}

// dw32: fp32 (Cout,32,1,1) gradient of the padded 1x1 weights -> dw OIHW (Cout,3,3,3) of dw_dtype
int yolo_stem_unpack_wgrad(const float* dw32, int Cout, void* dw, int dw_dtype, hipStream_t st) {
    int grid = (Cout * 27 + 255) / 256;
    switch (dw_dtype) {
        case YOLO_F32:  hipLaunchKernelGGL((k_stem_unpack_dw<float>), dim3(grid), dim3(256), 0, st, dw32, Cout, (float*)dw); break;
        case YOLO_BF16: hipLaunchKernelGGL((k_stem_unpack_dw<bf16_t>), dim3(grid), dim3(256), 0, st, dw32, Cout, (bf16_t*)dw); break;
        case YOLO_F16:  hipLaunchKernelGGL((k_stem_unpack_dw<f16_t>), dim3(grid), dim3(256), 0, st, dw32, Cout, (f16_t*)dw); break;
        default: return YOLO_ERR_DTYPE;
    }
    return YOLO_LAUNCH_CHECK();
}

// 1 if yolo_stem_conv_fwd handles these arguments (fp32 NCHW image, bf16/f16 compute, Cout in {16,32,48,64,96,128})
int yolo_stem_conv_eligible(int img_dtype, int dtype, int Cout) {
    const int ct = Cout / 16;
    return img_dtype == YOLO_F32 && (dtype == YOLO_BF16 || dtype == YOLO_F16) && Cout % 16 == 0 &&
           (ct == 1 || ct == 2 || ct == 3 || ct == 4 || ct == 6 || ct == 8);
}

// y[N][OH][OW][ld >= Cout] (dtype) = conv3x3 stride 2 pad 1 of img (N,3,H,W) fp32 NCHW with wp = yolo_stem_pack_weights
// output ([Cout][32], dtype); stats: optional [8][2][Cout] BatchNorm accumulator (sum, sum of squares of the stored values).
// bias (optional fp32 [Cout]) and act (0 identity, 1 SiLU): the fused-inference form act(conv + bias) of Model.fuse();
// not combined with stats (the statistics are those of the raw conv output).
int yolo_stem_conv_fwd(const float* img, const void* wp, void* y, int ldy, float* stats, int N, int H, int W, int OH, int OW,
                       int Cout, const float* bias, int act, int dtype, hipStream_t st) {
    if ((bias != nullptr || act) && stats != nullptr) return YOLO_ERR_ARG;
    if (act != 0 && act != 1) return YOLO_ERR_ARG;
    if (!yolo_stem_conv_eligible(YOLO_F32, dtype, Cout) || ldy < Cout) return YOLO_ERR_ARG;
    if (OH != (H - 1) / 2 + 1 || OW != (W - 1) / 2 + 1) return YOLO_ERR_ARG;
    if (dtype == YOLO_BF16) return launch_stem_conv<bf16_t>(img, wp, y, ldy, stats, N, H, W, OH, OW, Cout, bias, act, st);
    return launch_stem_conv<f16_t>(img, wp, y, ldy, stats, N, H, W, OH, OW, Cout, bias, act, st);
}

// number of partial matrices yolo_stem_wgrad may write: partial = fp32 [yolo_stem_wgrad_slabs()][Cout][32] scratch
int yolo_stem_wgrad_slabs(void) { return STEM_WG_SLABS; }

// dw OIHW (Cout,3,3,3) of dw_dtype = weight gradient of the stem conv from the fp32 NCHW image and dy[N][OH][OW][ldy >= Cout]
// (dtype bf16/f16; same eligibility as yolo_stem_conv_fwd).  Replaces unfold + 1x1 weight gradient + unpack.
int yolo_stem_wgrad(const float* img, const void* dy, int ldy, float* partial, void* dw, int dw_dtype, int N, int H, int W,
                    int OH, int OW, int Cout, int dtype, hipStream_t st) {
    if (!yolo_stem_conv_eligible(YOLO_F32, dtype, Cout) || ldy < Cout || (ldy & 7) || (reinterpret_cast<uintptr_t>(dy) & 15)) return YOLO_ERR_ARG;
    if (OH != (H - 1) / 2 + 1 || OW != (W - 1) / 2 + 1) return YOLO_ERR_ARG;
    int nslab = 0;
    const int rc = dtype == YOLO_BF16 ? launch_stem_wgrad<bf16_t>(img, dy, ldy, partial, N, H, W, OH, OW, Cout, &nslab, st)
                                      : launch_stem_wgrad<f16_t>(img, dy, ldy, partial, N, H, W, OH, OW, Cout, &nslab, st);
    if (rc) return rc;
    const int grid = (Cout * 27 + 7) / 8;
    switch (dw_dtype) {
        case YOLO_F32:  hipLaunchKernelGGL((k_stem_wgrad_reduce<float>), dim3(grid), dim3(256), 0, st, partial, nslab, Cout, (float*)dw); break;
        case YOLO_BF16: hipLaunchKernelGGL((k_stem_wgrad_reduce<bf16_t>), dim3(grid), dim3(256), 0, st, partial, nslab, Cout, (bf16_t*)dw); break;
        case YOLO_F16:  hipLaunchKernelGGL((k_stem_wgrad_reduce<f16_t>), dim3(grid), dim3(256), 0, st, partial, nslab, Cout, (f16_t*)dw); break;
        default: return YOLO_ERR_DTYPE;
    }
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
