// YoloDFLQFLoss forward + gradient in one pass over preds (reference src/model/losses.py:140-281).
//
//   preds  [N][64+nc][A]  (T, anchor index fastest)  ->  out[3] = {total, mean_dfl, mean_cls} (fp32)
//                                                        dpreds [N][64+nc][A] (T) = d total / d preds
// THREE launches (round 3; six before: decode, assign, zero fill, dense, matched, finish):
//   k_front    : heterogeneous grid.  Blocks [0, DB): per (n,a) softmax-expectation of the 4x16 DFL logits -> centre-xywh
//                pixels (fp32 pbox).  Blocks [DB, DB+nblk): quality-focal term with an all-zero target for every class logit
//                (+ its gradient), zero gradient for the 64 box logits; per-block partial sums (deterministic).  The two
//                halves are independent and share the launch (and the chip) instead of two launches back to back.
//   k_assign   : one wave per GT: argmin over the image's anchors of the distance to the predicted centre, evaluated the way
//                torch.cdist does for >25 columns ( |a|^2 + |b|^2 - 2ab via a k-ordered fma chain, clamp, sqrt ) so
//                near-ties resolve as in the reference; first minimum wins.
//   k_back     : one block per image.  Per matched anchor: DFL cross-entropy pair and its gradient, the quirk IoU (b1_y2 =
//                h + cy/2, losses.py:20), the soft target at (anchor, class) of the LAST GT mapped there (losses.py:261), and
//                the IoU gradient that autograd hands to EVERY GT (also those that lost the slot) back through box decode
//                into the box logits.  The block that finishes LAST (a ticket counter, cleared by k_front) turns the
//                partials into the three scalars, summing in a fixed order.
// fp32 arithmetic, no fma contraction in this file (built with -ffp-contract=off) so the box decode
// follows the reference's operation order.
#include "common.h"

namespace {

constexpr int REG = 16;
constexpr float QEPS = 1e-12f;

struct LossDims {
    int N, A, nc;
    float lambda_dfl, lambda_cls;
};

template <typename T>
__device__ __forceinline__ void decode_block(const LossDims& d, int block, const T* __restrict__ preds, const T* __restrict__ anchors,
                                             const T* __restrict__ strides, float4* __restrict__ pbox) {
    long i = block * (long)blockDim.x + threadIdx.x;
    if (i >= (long)d.N * d.A) return;
    int a = (int)(i % d.A);
    long n = i / d.A;
    const T* p = preds + n * (long)(4 * REG + d.nc) * d.A + a;
    float e[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float x[REG], mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < REG; ++b) { x[b] = to_f<T>(p[(long)(s * REG + b) * d.A]); mx = fmaxf(mx, x[b]); }
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < REG; ++b) { x[b] = expf(x[b] - mx); sum += x[b]; }
        float ex = 0.f;
#pragma unroll
        for (int b = 0; b < REG; ++b) ex += (x[b] / sum) * (float)b;
        e[s] = ex;
    }
    float ax = to_f<T>(anchors[a]), ay = to_f<T>(anchors[d.A + a]), st = to_f<T>(strides[a]);
    float x1 = (ax - e[0]) * st, y1 = (ay - e[1]) * st, x2 = (ax + e[2]) * st, y2 = (ay + e[3]) * st;
    pbox[i] = make_float4((x1 + x2) / 2.f, (y1 + y2) / 2.f, x2 - x1, y2 - y1);
}

// assignment of GT row j (of image n) by ONE wave: argmin over the image's anchors, lowest index among equal distances,
// NaN wins like torch.argmin; every lane returns the winner
__device__ __forceinline__ int assign_wave(int A, const float4* __restrict__ pb, const float* __restrict__ gt, int j, int lane) {
    const float gx = gt[j * 5 + 0], gy = gt[j * 5 + 1];
    const float an = __fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy));
    const float m2x = -2.f * gx, m2y = -2.f * gy;
    float best = INFINITY;
    int bi = 0x7fffffff;
    auto consider = [&](int a, const float4& b) {
        float bn = __fadd_rn(__fmul_rn(b.x, b.x), __fmul_rn(b.y, b.y));
        float acc = __fmul_rn(m2x, b.x);
        acc = __fmaf_rn(m2y, b.y, acc);
        acc = __fadd_rn(an, acc);
        acc = __fadd_rn(bn, acc);
        float dist = __fsqrt_rn(fmaxf(acc, 0.f));
        if (dist < best || (dist != dist && best == best)) { best = dist; bi = a; }   // NaN wins like argmin
    };
    // eight boxes per lane in flight, considered in index order
    int a = lane;
    for (; a + 7 * 64 < A; a += 8 * 64) {
        float4 b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = pb[a + u * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) consider(a + u * 64, b[u]);
    }
    for (; a < A; a += 64) consider(a, pb[a]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        bool take = (ob < best) || (ob == best && oi < bi) || (ob != ob && (best == best || oi < bi));
        if (take) { best = ob; bi = oi; }
    }
    return bi;
}

// grid = G rows of the gt buffer, one wave per row (four rows per block); rows past the live count gt_off[N] (a buffer with
// spare capacity, refilled between replays of a captured step) do nothing.  (Folding this into k_back -- each image's block
// assigning its own GTs one after the other -- was measured: 146 us for the loss instead of 117; the rows need the chip's
// parallelism, not one block per image.)
__global__ __launch_bounds__(256) void k_assign(int A, const float4* __restrict__ pbox, const float* __restrict__ gt,
                                                const int* __restrict__ gt_img, const int* __restrict__ gt_off, int N,
                                                int* __restrict__ idx) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= gt_off[N]) return;
    const int bi = assign_wave(A, pbox + (long)gt_img[j] * A, gt, j, lane);
    if (lane == 0) idx[j] = bi;
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// QFL element with target t: value and d/dlogit (losses.py:51-55)
__device__ __forceinline__ void qfl_elem(float x, float t, float& val, float& dx) {
    float s = sigm(x);
    float ls = logf(s + QEPS), l1 = logf(1.f - s + QEPS);
    float om = 1.f - s;
    val = -(t * om * om * ls + (1.f - t) * s * s * l1);
    float dpos = -2.f * om * ls + om * om / (s + QEPS);
    float dneg = 2.f * s * l1 - s * s / (1.f - s + QEPS);
    dx = -(t * dpos + (1.f - t) * dneg) * s * om;
}

// the same with target 0 (every class logit of the dense pass): t * (...) vanishes exactly (every factor is finite), so
// dropping the positive branch -- a logf and a division per element -- leaves value and gradient bit-identical
__device__ __forceinline__ void qfl_elem0(float x, float& val, float& dx) {
    float s = sigm(x);
    float l1 = logf(1.f - s + QEPS);
    float om = 1.f - s;
    val = -(s * s * l1);
    float dneg = 2.f * s * l1 - s * s / (1.f - s + QEPS);
    dx = -dneg * s * om;
}

// Per image the class logits (nc x A) and the box logits (64 x A) are two contiguous regions: the class packets of all
// images are spread evenly over the grid, four per thread in flight (one 32-bit division per packet finds its image);
// the box region only gets its zero gradient written.
template <typename T, int V>
__device__ __forceinline__ void dense_block(const LossDims& d, int block, int nblk, const T* __restrict__ preds, T* __restrict__ dpreds,
                                            float coef, double* __restrict__ partial, const float* __restrict__ grad_scale) {
    __shared__ float red[4];
    if (grad_scale) coef *= *grad_scale;                     // fp16 loss scaling: applied in fp32, before the one rounding
    const int Cp = 4 * REG + d.nc;
    const int P = d.nc * (d.A / V), Z = 4 * REG * (d.A / V);  // packets per image: class region, box region (host: A % V == 0)
    const int nthr = nblk * 256, gtid = block * 256 + threadIdx.x;
    constexpr int U = 4;
    float lsum = 0.f;
    const long per_img = (long)Cp * d.A;
    for (long i0 = gtid; i0 < (long)d.N * P; i0 += (long)nthr * U) {
        pack_t<T, V> raw[U];
        long eo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long idx = i0 + (long)u * nthr;
            const long ic = idx < (long)d.N * P ? idx : i0;
            const int n = (int)(ic / P), off = (int)(ic - (long)n * P);
            eo[u] = n * per_img + (long)4 * REG * d.A + (long)off * V;
            raw[u] = load_raw<T, V>(preds + eo[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + (long)u * nthr >= (long)d.N * P) break;
            float x[V], g[V];
            unpack<T, V>(raw[u], x);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float v, dx;
                qfl_elem0(x[e], v, dx);
                lsum += v;
                g[e] = dx * coef;
            }
            if (dpreds) store_pack<T, V>(dpreds + eo[u], g);
        }
    }
    if (dpreds) {
        float z[V];
#pragma unroll
        for (int k = 0; k < V; ++k) z[k] = 0.f;
        for (long idx = gtid; idx < (long)d.N * Z; idx += nthr) {
            const int n = (int)(idx / Z), off = (int)(idx - (long)n * Z);
            store_pack<T, V>(dpreds + n * per_img + (long)off * V, z);
        }
    }
    lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) partial[block] = (double)red[0] + red[1] + red[2] + red[3];
}

// blocks [0, DB): box decode; blocks [DB, DB + nblk): the dense class pass.  Block 0 also clears the ticket counter k_back's
// last block is found with.
template <typename T, int V>
__global__ __launch_bounds__(256) void k_front(LossDims d, int DB, int nblk, const T* __restrict__ preds, const T* __restrict__ anchors,
                                               const T* __restrict__ strides, float4* __restrict__ pbox, T* __restrict__ dpreds,
                                               float coef, double* __restrict__ partial, const float* __restrict__ grad_scale,
                                               unsigned int* __restrict__ ticket) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;
    if ((int)blockIdx.x < DB) decode_block<T>(d, blockIdx.x, preds, anchors, strides, pbox);
    else dense_block<T, V>(d, (int)blockIdx.x - DB, nblk, preds, dpreds, coef, partial, grad_scale);
}

__device__ void finish_scalars(const LossDims& d, const double* partial, int nblk, const double* img_dfl, const double* img_cls_fix,
                               float* __restrict__ out);

// one workgroup per image; wave w takes GTs w, w+4, ...; lane = (side, bin) of the 64 box logits
template <typename T>
__global__ __launch_bounds__(256) void k_back(LossDims d, const T* __restrict__ preds, T* __restrict__ dpreds,
                                              const T* __restrict__ anchors, const T* __restrict__ strides,
                                              const float4* __restrict__ pbox, const float* __restrict__ gt,
                                              const int* __restrict__ gt_off, const int* __restrict__ idx,
                                              double* __restrict__ img_dfl, double* __restrict__ img_cls_fix,
                                              const float* __restrict__ grad_scale, const double* __restrict__ partial, int nblk,
                                              unsigned int* __restrict__ ticket, float* __restrict__ out) {
    __shared__ double w_dfl[4], w_cls[4];
    __shared__ unsigned int my_ticket;
    const float gsc = grad_scale ? *grad_scale : 1.f;
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g0 = gt_off[n], M = gt_off[n + 1] - g0;

    const int Cp = 4 * REG + d.nc;
    const T* pn = preds + (long)n * Cp * d.A;
    T* dn = dpreds ? dpreds + (long)n * Cp * d.A : nullptr;
    const int side = lane >> 4, bin = lane & 15;
    const float cdfl = d.lambda_dfl / (float)d.N * 0.25f / (float)(M > 0 ? M : 1);
    const float ccls = d.lambda_cls / (float)d.N / (float)d.A;
    double acc_dfl = 0.0, acc_cls = 0.0;

    for (int j = wave; j < M; j += 4) {
        const int a = idx[g0 + j];
        bool owner = true;
        for (int q = 0; q < j; ++q) owner &= (idx[g0 + q] != a);
        if (!owner) continue;                                   // wave-uniform
        // softmax of this lane's side over its 16 bins
        float x = to_f<T>(pn[(long)lane * d.A + a]);
        float mx = x;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float ex = expf(x - mx), sum = ex;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        float p = ex / sum, lse = mx + logf(sum);
        float ev = p * (float)bin;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ev += __shfl_xor(ev, o, 64);           // expectation of my side
        const float ax = to_f<T>(anchors[a]), ay = to_f<T>(anchors[d.A + a]), st = to_f<T>(strides[a]);
        const float4 pb = pbox[(long)n * d.A + a];
        float gacc = 0.f;                                        // d total / d logit(lane)
        int winner = -1;
        for (int q = j; q < M; ++q)
            if (idx[g0 + q] == a) winner = q;
        for (int q = j; q < M; ++q) {
            if (idx[g0 + q] != a) continue;
            const float* gq = gt + (long)(g0 + q) * 5;
            const float gx = gq[0], gy = gq[1], gw = gq[2], gh = gq[3];
            const int cls = (int)gq[4];
            const float gx1 = gx - gw / 2.f, gy1 = gy - gh / 2.f, gx2 = gx + gw / 2.f, gy2 = gy + gh / 2.f;
            // ---- DFL (losses.py:226-253)
            float tl = ax - gx1 / st, tt = ay - gy1 / st, tr = gx2 / st - ax, tb = gy2 / st - ay;
            float tv = side == 0 ? tl : side == 1 ? tt : side == 2 ? tr : tb;
            tv = fminf(fmaxf(tv, 0.f), (float)(REG - 1) - 0.01f);
            int lo = (int)tv;
            float wl = (float)(lo + 1) - tv, wr = tv - (float)lo;
            float contrib = (bin == lo ? wl * (lse - x) : 0.f) + (bin == lo + 1 ? wr * (lse - x) : 0.f);
            float side_loss = contrib;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) side_loss += __shfl_xor(side_loss, o, 64);
            float all_sides = side_loss + __shfl_xor(side_loss, 16, 64);
            all_sides += __shfl_xor(all_sides, 32, 64);
            if (lane == 0) acc_dfl += (double)all_sides * 0.25 / (double)M;
            gacc += cdfl * (p - (bin == lo ? wl : 0.f) - (bin == lo + 1 ? wr : 0.f));
            // ---- quirk IoU of (pred box, GT) and its gradient (losses.py:17-40)
            const float X1 = pb.x - pb.z / 2.f, Y1 = pb.y - pb.w / 2.f, X2 = pb.x + pb.z / 2.f, Y2 = pb.w + pb.y / 2.f;
            const float iwr = fminf(X2, gx2) - fmaxf(X1, gx1), ihr = fminf(Y2, gy2) - fmaxf(Y1, gy1);
            const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
            const float inter = iw * ih;
            const float a1 = (X2 - X1) * (Y2 - Y1), a2 = (gx2 - gx1) * (gy2 - gy1);
            const float den = a1 + a2 - inter + 1e-6f;
            const float iou = inter / den;
            // class logit of this GT at the matched anchor
            const float xc = to_f<T>(pn[(long)(4 * REG + cls) * d.A + a]);
            const float sc = sigm(xc);
            const float omc = 1.f - sc;
            // d total / d iou  (QFL is linear in the target)
            const float giou = ccls * -(omc * omc * logf(sc + QEPS) - sc * sc * logf(1.f - sc + QEPS));
            // min/max split a tie evenly, clamp(min=0) passes gradient at >= 0 (ATen semantics)
            const float mX2 = X2 < gx2 ? 1.f : (X2 == gx2 ? 0.5f : 0.f), mX1 = X1 > gx1 ? 1.f : (X1 == gx1 ? 0.5f : 0.f);
            const float mY2 = Y2 < gy2 ? 1.f : (Y2 == gy2 ? 0.5f : 0.f), mY1 = Y1 > gy1 ? 1.f : (Y1 == gy1 ? 0.5f : 0.f);
            const float pw = iwr >= 0.f ? 1.f : 0.f, ph = ihr >= 0.f ? 1.f : 0.f;
            const float dI_X2 = ih * pw * mX2, dI_X1 = -ih * pw * mX1, dI_Y2 = iw * ph * mY2, dI_Y1 = -iw * ph * mY1;
            const float dA_X2 = (Y2 - Y1), dA_X1 = -(Y2 - Y1), dA_Y2 = (X2 - X1), dA_Y1 = -(X2 - X1);
            const float kI = 1.f / den + inter / (den * den), kA = inter / (den * den);
            const float dX1 = giou * (kI * dI_X1 - kA * dA_X1), dX2 = giou * (kI * dI_X2 - kA * dA_X2);
            const float dY1 = giou * (kI * dI_Y1 - kA * dA_Y1), dY2 = giou * (kI * dI_Y2 - kA * dA_Y2);
            const float dcx = dX1 + dX2, dw = (dX2 - dX1) / 2.f, dcy = dY1 + dY2 / 2.f, dh = dY2 - dY1 / 2.f;
            const float dx1 = dcx / 2.f - dw, dx2 = dcx / 2.f + dw, dy1 = dcy / 2.f - dh, dy2 = dcy / 2.f + dh;
            const float gs = side == 0 ? -st * dx1 : side == 1 ? -st * dy1 : side == 2 ? st * dx2 : st * dy2;
            gacc += gs * p * ((float)bin - ev);
            // ---- soft target fix-up at (a, cls) for the GT that owns the slot (last one wins)
            if (q == winner && lane == 0) {
                float v0, g0_, v1, g1_;
                qfl_elem(xc, 0.f, v0, g0_);
                qfl_elem(xc, iou, v1, g1_);
                acc_cls += (double)v1 - (double)v0;
                if (dn) dn[(long)(4 * REG + cls) * d.A + a] = from_f<T>(g1_ * ccls * gsc);
            }
        }
        if (dn) dn[(long)lane * d.A + a] = from_f<T>(gacc * gsc);
    }
    if (lane == 0) { w_dfl[wave] = acc_dfl; w_cls[wave] = acc_cls; }
    __syncthreads();
    if (threadIdx.x == 0) {
        img_dfl[n] = w_dfl[0] + w_dfl[1] + w_dfl[2] + w_dfl[3];
        img_cls_fix[n] = w_cls[0] + w_cls[1] + w_cls[2] + w_cls[3];
        __threadfence();                                        // the two sums before the ticket
        my_ticket = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    // ---- (3) the block whose ticket is the last one sums everything, in a fixed order (independent of which block it is)
    if (my_ticket != (unsigned)d.N - 1u || wave != 0) return;
    __threadfence();
    finish_scalars(d, partial, nblk, img_dfl, img_cls_fix, out);
}

// one wave: lanes sum strided slices (a single thread walking 2048 partials took 113 us), then a fixed-order butterfly --
// deterministic.  The per-image sums were written by other blocks of this launch: agent-scope atomic loads.
__device__ void finish_scalars(const LossDims& d, const double* partial, int nblk, const double* img_dfl, const double* img_cls_fix,
                               float* __restrict__ out) {
    double cls = 0.0, dfl = 0.0;
    {   // eight partials per lane in flight (one load per trip was one round trip per trip: 32 of them, 10.8 us); the sums are
        // taken in the same order as before
        int b = threadIdx.x;
        for (; b + 7 * 64 < nblk; b += 8 * 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[b + u * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) cls += v[u];
        }
        for (; b < nblk; b += 64) cls += partial[b];
    }
    for (int n = threadIdx.x; n < d.N; n += 64) {
        cls += __hip_atomic_load(img_cls_fix + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dfl += __hip_atomic_load(img_dfl + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cls += __shfl_xor(cls, o, 64);
        dfl += __shfl_xor(dfl, o, 64);
    }
    if (threadIdx.x) return;
    double mean_cls = cls / (double)d.A / (double)d.N, mean_dfl = dfl / (double)d.N;
    out[0] = (float)(d.lambda_dfl * mean_dfl + d.lambda_cls * mean_cls);
    out[1] = (float)mean_dfl;
    out[2] = (float)mean_cls;
}

template <typename T>
__global__ void k_scale(T* __restrict__ x, long n, const float* __restrict__ s) {
    const float f = *s;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = from_f<T>(to_f<T>(x[i]) * f);
}

constexpr int DENSE_BLOCKS = 2048;

}  // namespace

extern "C" {

// workspace bytes: pbox (16 B per anchor) + idx (4 B per GT) + partials + the ticket counter
size_t yolo_loss_workspace_bytes(int N, int A, int G) {
    size_t b = (size_t)N * A * 16;
    b += ((size_t)(G > 0 ? G : 1) * 4 + 15) / 16 * 16;
    b += (size_t)(DENSE_BLOCKS + 2 * N) * 8;
    b += 16;
    return b;
}

// gt: fp32 [G][5] (cx,cy,w,h,cls, pixels) grouped by image; gt_off: int32 [N+1]; gt_img: int32 [G].  G is the number of
// ROWS of gt / gt_img; the live count is gt_off[N] <= G, read on the device (a captured step refills the buffers).
// dpreds may be null (validation).  out: fp32[3] = {total, mean_dfl ("box_loss"), mean_cls} (never scaled).
// grad_scale: optional DEVICE scalar (fp16 loss scaling, train_model.py:247-253): dpreds = scale * d total / d preds, the
// factor applied in fp32 before the gradient is rounded to T -- what `scaler.scale(loss).backward()` computes in the
// reference, where the scaled loss gradient is formed in fp32 and cast at the `.float()` boundary (losses.py:142).
int yolo_loss_dfl_qfl(const void* preds, const void* anchors, const void* strides, int dtype, int N, int nc, int A,
                      const float* gt, const int* gt_off, const int* gt_img, int G, float lambda_dfl,
                      float lambda_cls, void* dpreds, float* out, void* workspace, const float* grad_scale, hipStream_t st) {
    if (N < 1) return YOLO_ERR_ARG;
    LossDims d{N, A, nc, lambda_dfl, lambda_cls};
    char* ws = (char*)workspace;
    float4* pbox = (float4*)ws;
    ws += (size_t)N * A * 16;
    int* idx = (int*)ws;
    ws += ((size_t)(G > 0 ? G : 1) * 4 + 15) / 16 * 16;
    double* partial = (double*)ws;
    double* img_dfl = partial + DENSE_BLOCKS;
    double* img_fix = img_dfl + N;
    unsigned int* ticket = (unsigned int*)(img_fix + N);
    const long NA = (long)N * A;
    const long elems = NA * (64 + nc);
    const float coef = lambda_cls / (float)N / (float)A;
    const int DB = ceil_div(NA, 256);
    YOLO_DISPATCH_T(dtype, {
        constexpr int VV = vec_of<T>::N;
        bool vec = (A % VV == 0) && ((uintptr_t)preds % 16 == 0) && (!dpreds || (uintptr_t)dpreds % 16 == 0);
        int nblk = (int)((elems / (vec ? VV : 1) + 255) / 256);
        if (nblk > DENSE_BLOCKS) nblk = DENSE_BLOCKS;
        if (nblk < 1) nblk = 1;
        if (vec) hipLaunchKernelGGL((k_front<T, VV>), dim3(DB + nblk), dim3(256), 0, st, d, DB, nblk, (const T*)preds, (const T*)anchors,
                                    (const T*)strides, pbox, (T*)dpreds, coef, partial, grad_scale, ticket);
        else hipLaunchKernelGGL((k_front<T, 1>), dim3(DB + nblk), dim3(256), 0, st, d, DB, nblk, (const T*)preds, (const T*)anchors,
                                (const T*)strides, pbox, (T*)dpreds, coef, partial, grad_scale, ticket);
        if (G > 0) hipLaunchKernelGGL(k_assign, dim3(ceil_div(G, 4)), dim3(256), 0, st, A, pbox, gt, gt_img, gt_off, N, idx);
        hipLaunchKernelGGL((k_back<T>), dim3(N), dim3(256), 0, st, d, (const T*)preds, (T*)dpreds, (const T*)anchors,
                           (const T*)strides, pbox, gt, gt_off, idx, img_dfl, img_fix, grad_scale, partial, nblk, ticket, out);
    });
    return YOLO_LAUNCH_CHECK();
}

// x *= *scale_dev (scale read on the device: no host sync); used to apply the incoming grad_output
int yolo_scale_inplace(void* x, long n, int dtype, const float* scale_dev, hipStream_t st) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    YOLO_DISPATCH_T(dtype, hipLaunchKernelGGL((k_scale<T>), dim3((int)b), dim3(256), 0, st, (T*)x, n, scale_dev));
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
