// Inference post-processing: head decode (reference src/model/model_builder.py:123-136) and
// class-aware NMS (src/utils/model_utils.py:174-279 + torchvision.ops.nms semantics).
//
// NMS, per image, all on device, deterministic:
//   k_candidates : confidence filter + best class (or every class above the threshold when
//                  multi_label), optional class whitelist, xywh->xyxy with each operation rounded in
//                  the input dtype T (the reference does this arithmetic in T before torch.cat promotes
//                  the rows to fp32); order-preserving compaction by block scans
//   k_rank       : stable descending order by counting (rank = #better), cap 30000 (max_nms)
//   k_mask       : 64x64 tiles of "IoU(i,j) > thr, j after i" bits on boxes + cls*7680 (class-aware trick
//                  kept verbatim so the low bits of the fp32 coordinates match the reference)
//   k_scan       : one wave walks the sorted list, keeps at most max_det
// fp32 IoU exactly as torchvision: inter / (area_i + area_j - inter), strict >, no eps.
// Built with -ffp-contract=off: bit-exact index selection needs the un-fused multiply/add order.
#include "common.h"

namespace {

constexpr int REG = 16;
constexpr float MAX_WH = 7680.f;
constexpr int MAX_NMS = 30000;
constexpr int NMS_CHUNK = 8;         // images per mask / scan launch (each owns a ns x words bit matrix: 112 MB at 30000 candidates)

template <typename T> __device__ __forceinline__ float rt(float v) { return to_f<T>(from_f<T>(v)); }

template <typename T>
__global__ void k_head_decode(const T* __restrict__ preds, const T* __restrict__ anchors, const T* __restrict__ strides,
                              T* __restrict__ y, int N, int nc, int A) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)N * A) return;
    int a = (int)(i % A);
    long n = i / A;
    const T* p = preds + n * (long)(4 * REG + nc) * A + a;
    T* o = y + n * (long)(4 + nc) * A + a;
    float e[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float x[REG], mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < REG; ++b) { x[b] = to_f<T>(p[(long)(s * REG + b) * A]); mx = fmaxf(mx, x[b]); }
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < REG; ++b) { x[b] = expf(x[b] - mx); sum += x[b]; }
        float ex = 0.f;
#pragma unroll
        for (int b = 0; b < REG; ++b) ex += (x[b] / sum) * (float)b;
        e[s] = ex;
    }
    float ax = to_f<T>(anchors[a]), ay = to_f<T>(anchors[A + a]), st = to_f<T>(strides[a]);
    float x1 = ax - e[0], y1 = ay - e[1], x2 = ax + e[2], y2 = ay + e[3];
    o[0] = from_f<T>((x1 + x2) / 2.f * st);
    o[(long)A] = from_f<T>((y1 + y2) / 2.f * st);
    o[2L * A] = from_f<T>((x2 - x1) * st);
    o[3L * A] = from_f<T>((y2 - y1) * st);
    for (int c = 0; c < nc; ++c) o[(long)(4 + c) * A] = p[(long)(4 * REG + c) * A];
}

// DFL block alone (src/model/model_blocks.py:278-280): (b, 64, a) logits -> (b, 4, a) expected distances
template <typename T>
__global__ void k_dfl_expect(const T* __restrict__ x, T* __restrict__ y, int B, int A) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * 4 * A) return;
    int a = (int)(i % A);
    long t = i / A;
    int s = (int)(t % 4);
    long b = t / 4;
    const T* p = x + (b * 64 + s * REG) * (long)A + a;
    float v[REG], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < REG; ++k) { v[k] = to_f<T>(p[(long)k * A]); mx = fmaxf(mx, v[k]); }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < REG; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
    float ex = 0.f;
#pragma unroll
    for (int k = 0; k < REG; ++k) ex += (v[k] / sum) * (float)k;
    y[i] = from_f<T>(ex);
}

struct ClsFilter { int n; int ids[32]; };

__device__ __forceinline__ bool cls_ok(const ClsFilter& f, int c) {
    if (f.n == 0) return true;
    bool ok = false;
    for (int k = 0; k < f.n; ++k) ok |= (f.ids[k] == c);
    return ok;
}

// block-wide exclusive scan of one int per thread (256 threads); returns the block total in `total`
__device__ __forceinline__ int block_excl_scan(int v, int* lds4, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds4[w];
    total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    return base + inc - v;
}

// candidates of anchor m: their number (0 / 1, or the passing classes with multi_label), best class and score
template <typename T>
__device__ __forceinline__ int cand_eval(const T* __restrict__ yi, int m, int M, int nc, int multi_label, float conf_thres,
                                         const ClsFilter& filt, float& best, int& bestc) {
    int cnt = 0;
    best = -INFINITY;
    bestc = 0;
    if (m < M) {
        for (int c = 0; c < nc; ++c) {
            float v = to_f<T>(yi[(long)(4 + c) * M + m]);
            if (multi_label) { if (v > conf_thres && cls_ok(filt, c)) ++cnt; }
            else if (v > best) { best = v; bestc = c; }
        }
        if (!multi_label) cnt = (best > conf_thres && cls_ok(filt, bestc)) ? 1 : 0;
    }
    return cnt;
}

// Order-preserving compaction of the candidates in two passes over the anchors, one workgroup per 256 anchors and
// image (the first form walked an image's anchors with ONE workgroup: 3.1 ms for 8 images x 33600 anchors x 84 rows,
// 80 % of the NMS): pass 1 counts per chunk, pass 2 places each chunk behind the sum of the chunks before it.
template <typename T>
__global__ __launch_bounds__(256) void k_cand_count(const T* __restrict__ y, int nc, int M, float conf_thres, int multi_label,
                                                    ClsFilter filt, int nchunk, int* __restrict__ chunk_cnt /*[bs][nchunk]*/) {
    __shared__ int lds4[4];
    const int img = blockIdx.y;
    const T* yi = y + (long)img * (4 + nc) * M;
    conf_thres = rt<T>(conf_thres);      // torch compares a T tensor with the Python scalar cast to T
    float best;
    int bestc, total;
    const int cnt = cand_eval<T>(yi, blockIdx.x * 256 + threadIdx.x, M, nc, multi_label, conf_thres, filt, best, bestc);
    block_excl_scan(cnt, lds4, total);
    if (threadIdx.x == 0) chunk_cnt[(long)img * nchunk + blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(256) void k_candidates(const T* __restrict__ y, int nc, int M, float conf_thres,
                                                    int multi_label, ClsFilter filt, int cap, int nchunk,
                                                    const int* __restrict__ chunk_cnt,
                                                    float* __restrict__ rows /*[bs][cap][6]*/,
                                                    int* __restrict__ count /*[bs]*/, int* __restrict__ overflow) {
    __shared__ int lds4[4];
    __shared__ int sbase;
    const int img = blockIdx.y;
    const T* yi = y + (long)img * (4 + nc) * M;
    float* out = rows + (long)img * cap * 6;
    conf_thres = rt<T>(conf_thres);
    {   // candidates of the chunks before this one
        int part = 0;
        for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) part += chunk_cnt[(long)img * nchunk + c];
        int tot;
        block_excl_scan(part, lds4, tot);
        if (threadIdx.x == 0) sbase = tot;
        __syncthreads();
    }
    const int base = sbase;
    const int m = blockIdx.x * 256 + threadIdx.x;
    float best;
    int bestc, total;
    const int cnt = cand_eval<T>(yi, m, M, nc, multi_label, conf_thres, filt, best, bestc);
    int pos = base + block_excl_scan(cnt, lds4, total);
    if (cnt > 0) {
        float cx = to_f<T>(yi[m]), cy = to_f<T>(yi[(long)M + m]);
        float dw = rt<T>(to_f<T>(yi[2L * M + m]) / 2.f), dh = rt<T>(to_f<T>(yi[3L * M + m]) / 2.f);
        float x1 = rt<T>(cx - dw), y1 = rt<T>(cy - dh), x2 = rt<T>(cx + dw), y2 = rt<T>(cy + dh);
        if (!multi_label) {
            if (pos < cap) {
                float* r = out + (long)pos * 6;
                r[0] = x1; r[1] = y1; r[2] = x2; r[3] = y2; r[4] = best; r[5] = (float)bestc;
            }
        } else {
            for (int c = 0; c < nc; ++c) {
                float v = to_f<T>(yi[(long)(4 + c) * M + m]);
                if (v > conf_thres && cls_ok(filt, c)) {
                    if (pos < cap) {
                        float* r = out + (long)pos * 6;
                        r[0] = x1; r[1] = y1; r[2] = x2; r[3] = y2; r[4] = v; r[5] = (float)c;
                    }
                    ++pos;
                }
            }
        }
    }
    if (blockIdx.x == (unsigned)nchunk - 1 && threadIdx.x == 0) {
        int n = base + total;
        if (n > cap) { *overflow = 1; n = cap; }
        count[img] = n;
    }
}

// rank by counting; order[rank] = i for rank < MAX_NMS.   grid.x covers cap, grid.y = image
__global__ __launch_bounds__(256) void k_rank(const float* __restrict__ rows, const int* __restrict__ count, int cap,
                                              int* __restrict__ order /*[bs][MAX_NMS]*/) {
    __shared__ float sc[256];
    const int img = blockIdx.y;
    const int n = count[img];
    if (blockIdx.x * 256 >= n) return;
    const float* r = rows + (long)img * cap * 6;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float mine = i < n ? r[(long)i * 6 + 4] : 0.f;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        int j = j0 + threadIdx.x;
        sc[threadIdx.x] = j < n ? r[(long)j * 6 + 4] : -INFINITY;
        __syncthreads();
        int lim = n - j0 < 256 ? n - j0 : 256;
        for (int k = 0; k < lim; ++k) {
            float o = sc[k];
            rank += (o > mine) || (o == mine && j0 + k < i);
        }
    }
    if (i < n && rank < MAX_NMS) order[(long)img * MAX_NMS + rank] = i;
}

// mask[i][w] bit b: box order[w*64+b] is suppressed by box order[i]  (only j > i)
__global__ __launch_bounds__(64) void k_mask(const float* __restrict__ rows, const int* __restrict__ count,
                                             const int* __restrict__ order, int cap, int img0, float thr, int agnostic,
                                             int words, long mask_stride, unsigned long long* __restrict__ mask) {
    __shared__ float bx[64][4];
    const int img = img0 + blockIdx.z;                        // images of one chunk share the launch (own mask each)
    mask += (long)blockIdx.z * mask_stride;
    int n = count[img];
    if (n > MAX_NMS) n = MAX_NMS;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi || bi * 64 >= n || bj * 64 >= n) return;
    const float* r = rows + (long)img * cap * 6;
    const int* ord = order + (long)img * MAX_NMS;
    const int t = threadIdx.x;
    {
        int j = bj * 64 + t;
        if (j < n) {
            const float* q = r + (long)ord[j] * 6;
            float off = agnostic ? 0.f : q[5] * MAX_WH;
            bx[t][0] = q[0] + off; bx[t][1] = q[1] + off; bx[t][2] = q[2] + off; bx[t][3] = q[3] + off;
        }
    }
    __syncthreads();
    const int i = bi * 64 + t;
    if (i >= n) return;
    const float* q = r + (long)ord[i] * 6;
    const float off = agnostic ? 0.f : q[5] * MAX_WH;
    const float x1 = q[0] + off, y1 = q[1] + off, x2 = q[2] + off, y2 = q[3] + off;
    const float area = (x2 - x1) * (y2 - y1);
    unsigned long long bits = 0ull;
    const int lim = n - bj * 64 < 64 ? n - bj * 64 : 64;
    for (int k = 0; k < lim; ++k) {
        int j = bj * 64 + k;
        if (j <= i) continue;
        float w = fmaxf(0.f, fminf(x2, bx[k][2]) - fmaxf(x1, bx[k][0]));
        float h = fmaxf(0.f, fminf(y2, bx[k][3]) - fmaxf(y1, bx[k][1]));
        float inter = w * h;
        float aj = (bx[k][2] - bx[k][0]) * (bx[k][3] - bx[k][1]);
        float iou = inter / (area + aj - inter);
        if (iou > thr) bits |= 1ull << k;
    }
    mask[(long)i * words + bj] = bits;
}

__global__ __launch_bounds__(64) void k_scan(const float* __restrict__ rows, const int* __restrict__ count,
                                             const int* __restrict__ order, int cap, int img0, int words, int max_det,
                                             long mask_stride, const unsigned long long* __restrict__ mask,
                                             float* __restrict__ out /*[bs][max_det][6]*/, int* __restrict__ out_count) {
    extern __shared__ unsigned long long removed[];
    const int img = img0 + blockIdx.x;                        // one wave per image, the images of a chunk side by side
    mask += (long)blockIdx.x * mask_stride;
    int n = count[img];
    if (n > MAX_NMS) n = MAX_NMS;
    const int lane = threadIdx.x;
    const int nw = (n + 63) / 64;
    for (int w = lane; w < nw; w += 64) removed[w] = 0ull;
    __syncthreads();
    const float* r = rows + (long)img * cap * 6;
    const int* ord = order + (long)img * MAX_NMS;
    float* o = out + (long)img * max_det * 6;
    int kept = 0;
    for (int i = 0; i < n && kept < max_det; ++i) {
        unsigned long long rw = removed[i >> 6];
        if ((rw >> (i & 63)) & 1ull) continue;
        if (lane < 6) o[(long)kept * 6 + lane] = r[(long)ord[i] * 6 + lane];
        ++kept;
        __syncthreads();
        for (int w = (i >> 6) + lane; w < nw; w += 64) removed[w] |= mask[(long)i * words + w];
        __syncthreads();
    }
    if (lane == 0) out_count[img] = kept;
}

// ---- validation path: reference src/training/train_model.py:14-142 (decode_predictions) and
// src/training/metrics.py:68-157 (DetectionMetrics.update) -------------------------------------------------------
// k_val_candidates : per anchor sigmoid of every class logit rounded in T (the reference applies .sigmoid() to the T
//                    tensor), first maximum, `>= conf` compared in T; order-preserving compaction
// k_val_rank       : only for images with more survivors than top_k: descending score, lower anchor first on ties
//                    (torch.topk leaves the order of equal scores unspecified)
// k_val_gather     : the kept rows [cx, cy, w, h, cls, score] per image
// k_val_match      : one wave per image walks the predictions in order; each takes the free same-class target of
//                    highest IoU (strict >, so the first index wins a tie and IoU 0 never matches), TP when that IoU
//                    (as double) >= threshold; int64 counters [total_pred, total_gt, tp, fp, fn, class_tp[nc],
//                    class_fp[nc], class_fn[nc], class_gt[nc]] updated with atomics
template <typename T>
__device__ __forceinline__ int val_eval(const T* __restrict__ yi, int m, int M, int nc, float conf, float& best, int& bestc) {
    best = -INFINITY;
    bestc = 0;
    if (m >= M) return 0;
    for (int c = 0; c < nc; ++c) {
        const float v = to_f<T>(yi[(long)(4 + c) * M + m]);
        const float sg = rt<T>(1.f / (1.f + expf(-v)));
        if (sg > best) { best = sg; bestc = c; }
    }
    return best >= conf ? 1 : 0;
}

// two passes over 256-anchor chunks like the NMS candidates (k_cand_count / k_candidates): count, then place each chunk
// behind the chunks before it (one workgroup per IMAGE walked 8400 anchors x 80 sigmoids alone)
template <typename T>
__global__ __launch_bounds__(256) void k_val_count(const T* __restrict__ y, int nc, int M, float conf, int nchunk,
                                                   int* __restrict__ chunk_cnt /*[N][nchunk]*/) {
    __shared__ int lds4[4];
    const int img = blockIdx.y;
    float best;
    int bestc, total;
    const int keep = val_eval<T>(y + (long)img * (4 + nc) * M, blockIdx.x * 256 + threadIdx.x, M, nc, rt<T>(conf), best, bestc);
    block_excl_scan(keep, lds4, total);
    if (threadIdx.x == 0) chunk_cnt[(long)img * nchunk + blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(256) void k_val_candidates(const T* __restrict__ y, int nc, int M, float conf, int nchunk,
                                                        const int* __restrict__ chunk_cnt,
                                                        float* __restrict__ rows /*[N][M][6]*/, int* __restrict__ count) {
    __shared__ int lds4[4];
    __shared__ int sbase;
    const int img = blockIdx.y;
    const T* yi = y + (long)img * (4 + nc) * M;
    float* out = rows + (long)img * M * 6;
    {
        int part = 0;
        for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) part += chunk_cnt[(long)img * nchunk + c];
        int tot;
        block_excl_scan(part, lds4, tot);
        if (threadIdx.x == 0) sbase = tot;
        __syncthreads();
    }
    const int base = sbase;
    const int m = blockIdx.x * 256 + threadIdx.x;
    float best;
    int bestc, total;
    const int keep = val_eval<T>(yi, m, M, nc, rt<T>(conf), best, bestc);
    const int pos = base + block_excl_scan(keep, lds4, total);
    if (keep) {
        float* r = out + (long)pos * 6;
        r[0] = to_f<T>(yi[m]); r[1] = to_f<T>(yi[(long)M + m]); r[2] = to_f<T>(yi[2L * M + m]);
        r[3] = to_f<T>(yi[3L * M + m]); r[4] = (float)bestc; r[5] = best;
    }
    if (blockIdx.x == (unsigned)nchunk - 1 && threadIdx.x == 0) count[img] = base + total;
}

__global__ __launch_bounds__(256) void k_val_rank(const float* __restrict__ rows, const int* __restrict__ count, int M,
                                                  int top_k, int* __restrict__ sel /*[N][top_k]*/) {
    __shared__ float sc[256];
    const int img = blockIdx.y;
    const int n = count[img];
    if (n <= top_k || blockIdx.x * 256 >= n) return;
    const float* r = rows + (long)img * M * 6;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float mine = i < n ? r[(long)i * 6 + 5] : 0.f;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
        sc[threadIdx.x] = j < n ? r[(long)j * 6 + 5] : -INFINITY;
        __syncthreads();
        const int lim = n - j0 < 256 ? n - j0 : 256;
        for (int k = 0; k < lim; ++k) {
            const float o = sc[k];
            rank += (o > mine) || (o == mine && j0 + k < i);
        }
    }
    if (i < n && rank < top_k) sel[(long)img * top_k + rank] = i;
}

__global__ __launch_bounds__(64) void k_val_gather(const float* __restrict__ rows, const int* __restrict__ count, int M,
                                                   int top_k, const int* __restrict__ sel, float* __restrict__ out,
                                                   int* __restrict__ out_count) {
    const int img = blockIdx.x;
    const int n = count[img];
    const int k = n < top_k ? n : top_k;
    const float* r = rows + (long)img * M * 6;
    float* o = out + (long)img * top_k * 6;
    for (int e = threadIdx.x; e < top_k * 6; e += 64) {
        const int row = e / 6, col = e - row * 6;
        float v = 0.f;
        if (row < k) v = r[(long)(n > top_k ? sel[(long)img * top_k + row] : row) * 6 + col];
        o[e] = v;
    }
    if (threadIdx.x == 0) out_count[img] = k;
}

constexpr int VAL_MAX_T = 16;       // targets per lane: images with up to 1024 ground-truth boxes

struct Corners { float x1, y1, x2, y2; };
__device__ __forceinline__ Corners corners_of(const float* b) {     // metrics.py:19-24, each op rounded in fp32
    const float hw = b[2] / 2.f, hh = b[3] / 2.f;
    return {b[0] - hw, b[1] - hh, b[0] + hw, b[1] + hh};
}
__device__ __forceinline__ float iou_xywh(const Corners& a, const Corners& b) {     // metrics.py:29-41
    float w = fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), h = fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1);
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = w * h;
    const float a1 = (a.x2 - a.x1) * (a.y2 - a.y1), a2 = (b.x2 - b.x1) * (b.y2 - b.y1);
    return inter / ((a1 + a2) - inter + 1e-6f);
}
__device__ __forceinline__ void bump(long long* c, long long v) {
    if (v) atomicAdd(reinterpret_cast<unsigned long long*>(c), (unsigned long long)v);
}

__global__ __launch_bounds__(64) void k_val_match(const float* __restrict__ pred /*[N][top_k][6]*/,
                                                  const int* __restrict__ count, int top_k,
                                                  const float* __restrict__ gt /*[total][5]*/,
                                                  const int* __restrict__ gt_off, double thr, int nc, int skip_empty_gt,
                                                  long long* __restrict__ ctr, int* __restrict__ status) {
    const int img = blockIdx.x, lane = threadIdx.x;
    const int n = count[img];
    const int g0 = gt_off[img], m = gt_off[img + 1] - g0;
    if ((n == 0 && m == 0) || (m == 0 && skip_empty_gt)) return;
    if (m > 64 * VAL_MAX_T) { if (lane == 0) *status = 1; return; }
    long long *c_tp = ctr + 5, *c_fp = c_tp + nc, *c_fn = c_fp + nc, *c_gt = c_fn + nc;
    const float* P = pred + (long)img * top_k * 6;
    const float* G = gt + (long)g0 * 5;
    if (n == 0) {                                               // :91-98 (totals are not advanced on this path)
        for (int j = lane; j < m; j += 64) {
            const long long c = (long long)G[j * 5 + 4];
            if (c >= 0 && c < nc) { bump(c_fn + c, 1); bump(c_gt + c, 1); }
        }
        if (lane == 0) bump(ctr + 4, m);
        return;
    }
    if (m == 0) {                                               // :100-106
        for (int i = lane; i < n; i += 64) {
            const long long c = (long long)P[i * 6 + 4];
            if (c >= 0 && c < nc) bump(c_fp + c, 1);
        }
        if (lane == 0) bump(ctr + 3, n);
        return;
    }
    unsigned taken = 0;                                         // bit t: target lane + 64 t is matched
    int tp = 0;
    for (int i = 0; i < n; ++i) {
        const Corners pb = corners_of(P + i * 6);
        const long long pc = (long long)P[i * 6 + 4];
        float best = 0.f;
        int bj = 0x7fffffff;
        for (int t = 0, j = lane; j < m; ++t, j += 64) {
            if ((taken >> t) & 1u) continue;
            if ((long long)G[j * 5 + 4] != pc) continue;
            const float v = iou_xywh(pb, corners_of(G + j * 5));
            if (v > best) { best = v; bj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                      // max IoU, lowest target index on ties
            const float ob = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        const bool hit = bj != 0x7fffffff && (double)best >= thr;
        if (hit && (bj & 63) == lane) taken |= 1u << (bj >> 6);
        if (lane == 0) {
            tp += hit;
            if (pc >= 0 && pc < nc) bump((hit ? c_tp : c_fp) + pc, 1);
        }
    }
    int unmatched = 0;
    for (int t = 0, j = lane; j < m; ++t, j += 64) {            // :148-156
        const long long c = (long long)G[j * 5 + 4];
        const bool free_ = !((taken >> t) & 1u);
        unmatched += free_;
        if (c >= 0 && c < nc) { bump(c_gt + c, 1); if (free_) bump(c_fn + c, 1); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) unmatched += __shfl_xor(unmatched, o, 64);
    if (lane == 0) {
        bump(ctr + 0, n); bump(ctr + 1, m); bump(ctr + 2, tp); bump(ctr + 3, n - tp); bump(ctr + 4, unmatched);
    }
}

}  // namespace

extern "C" {

int yolo_head_decode(const void* preds, const void* anchors, const void* strides, void* y, int N, int nc, int A,
                     int dtype, hipStream_t st) {
    long NA = (long)N * A;
    YOLO_DISPATCH_T(dtype, hipLaunchKernelGGL((k_head_decode<T>), dim3(ceil_div(NA, 256)), dim3(256), 0, st, (const T*)preds,
                                              (const T*)anchors, (const T*)strides, (T*)y, N, nc, A));
    return YOLO_LAUNCH_CHECK();
}

int yolo_dfl_expect(const void* x, void* y, int B, int A, int dtype, hipStream_t st) {
    long tot = (long)B * 4 * A;
    YOLO_DISPATCH_T(dtype, hipLaunchKernelGGL((k_dfl_expect<T>), dim3(ceil_div(tot, 256)), dim3(256), 0, st, (const T*)x, (T*)y, B, A));
    return YOLO_LAUNCH_CHECK();
}

// candidate capacity per image used by yolo_nms and its workspace
int yolo_nms_capacity(int M, int nc, int multi_label) {
    long c = (long)M * (multi_label ? nc : 1);
    return (int)(c > 131072 ? 131072 : c);
}

size_t yolo_nms_workspace_bytes(int bs, int M, int nc, int multi_label) {
    size_t cap = (size_t)yolo_nms_capacity(M, nc, multi_label);
    size_t ns = cap < MAX_NMS ? cap : MAX_NMS;
    size_t words = (ns + 63) / 64;
    size_t b = (size_t)bs * cap * 6 * 4;            // rows
    b += (size_t)bs * MAX_NMS * 4;                  // order
    b += ((size_t)bs * 4 + 4 + 15) / 16 * 16;       // count[bs] + overflow
    b += ((size_t)bs * ((M + 255) / 256) * 4 + 15) / 16 * 16;   // candidates per 256-anchor chunk
    b = (b + 15) / 16 * 16;
    b += (size_t)(bs < NMS_CHUNK ? bs : NMS_CHUNK) * ns * words * 8;   // masks of the images that share a launch
    return b;
}

// y: (bs, 4+nc, M) of dtype.  out: fp32 [bs][max_det][6] rows (x1,y1,x2,y2,conf,cls); out_count: int32 [bs];
// status: int32[1], set to 1 if an image had more candidates than the capacity (result then invalid).
int yolo_nms(const void* y, int dtype, int bs, int nc, int M, float conf_thres, float iou_thres, const int* classes,
             int n_classes, int agnostic, int multi_label, int max_det, float* out, int* out_count, int* status,
             void* workspace, hipStream_t st) {
    if (n_classes > 32 || max_det < 1) return YOLO_ERR_ARG;
    const int cap = yolo_nms_capacity(M, nc, multi_label);
    const int ns = cap < MAX_NMS ? cap : MAX_NMS;
    const int words = (ns + 63) / 64;
    char* ws = (char*)workspace;
    float* rows = (float*)ws;
    ws += (size_t)bs * cap * 6 * 4;
    int* order = (int*)ws;
    ws += (size_t)bs * MAX_NMS * 4;
    int* count = (int*)ws;
    int* overflow = count + bs;
    ws += ((size_t)bs * 4 + 4 + 15) / 16 * 16;
    const int nchunk = (M + 255) / 256;
    int* chunk_cnt = (int*)ws;
    ws += ((size_t)bs * nchunk * 4 + 15) / 16 * 16;
    ws = (char*)(((uintptr_t)ws + 15) / 16 * 16);
    unsigned long long* mask = (unsigned long long*)ws;
    ClsFilter filt;
    filt.n = n_classes;
    for (int k = 0; k < n_classes; ++k) filt.ids[k] = classes[k];
    int rc = yolo_zero_async(overflow, 4, st);
    if (rc) return rc;
    YOLO_DISPATCH_T(dtype, {
        hipLaunchKernelGGL((k_cand_count<T>), dim3(nchunk, bs), dim3(256), 0, st, (const T*)y, nc, M, conf_thres, multi_label,
                           filt, nchunk, chunk_cnt);
        hipLaunchKernelGGL((k_candidates<T>), dim3(nchunk, bs), dim3(256), 0, st, (const T*)y, nc, M, conf_thres, multi_label,
                           filt, cap, nchunk, chunk_cnt, rows, count, overflow);
    });
    hipLaunchKernelGGL(k_rank, dim3(ceil_div(cap, 256), bs), dim3(256), 0, st, rows, count, cap, order);
    // up to NMS_CHUNK images per pair of launches: the serial one-wave scans of different images run side by side
    // (one image at a time: 0.67 ms per image on the config-5 tensor, 5.4 ms for its batch of 8)
    const long mask_stride = (long)ns * words;
    for (int img = 0; img < bs; img += NMS_CHUNK) {
        const int nimg = bs - img < NMS_CHUNK ? bs - img : NMS_CHUNK;
        hipLaunchKernelGGL(k_mask, dim3(words, words, nimg), dim3(64), 0, st, rows, count, order, cap, img, iou_thres, agnostic,
                           words, mask_stride, mask);
        hipLaunchKernelGGL(k_scan, dim3(nimg), dim3(64), (size_t)words * 8, st, rows, count, order, cap, img, words, max_det,
                           mask_stride, mask, out, out_count);
    }
    rc = hip_status(hipMemcpyAsync(status, overflow, 4, hipMemcpyDeviceToDevice, st));
    if (rc) return rc;
    return YOLO_LAUNCH_CHECK();
}

// ---- validation (train_model.py:14-142, metrics.py:68-157) ----
size_t yolo_val_workspace_bytes(int N, int M, int top_k) {
    return (size_t)N * M * 6 * 4 + (size_t)N * top_k * 4 + (size_t)N * 4 + 64 + (size_t)N * ((M + 255) / 256) * 4;
}

// y: decoded head output (N, 4+nc, M) of dtype (yolo_head_decode).  out: fp32 [N][top_k][6] rows
// (cx, cy, w, h, cls, score), zero padded; out_count: int32 [N].
int yolo_val_select(const void* y, int dtype, int N, int nc, int M, float conf, int top_k, float* out, int* out_count,
                    void* workspace, hipStream_t st) {
    if (N < 1 || M < 1 || top_k < 1 || nc < 1) return YOLO_ERR_ARG;
    char* ws = (char*)workspace;
    float* rows = (float*)ws;
    ws += (size_t)N * M * 6 * 4;
    int* sel = (int*)ws;
    ws += (size_t)N * top_k * 4;
    int* count = (int*)ws;
    ws += (size_t)N * 4 + 64;
    int* chunk_cnt = (int*)ws;
    const int nchunk = (M + 255) / 256;
    YOLO_DISPATCH_T(dtype, {
        hipLaunchKernelGGL((k_val_count<T>), dim3(nchunk, N), dim3(256), 0, st, (const T*)y, nc, M, conf, nchunk, chunk_cnt);
        hipLaunchKernelGGL((k_val_candidates<T>), dim3(nchunk, N), dim3(256), 0, st, (const T*)y, nc, M, conf, nchunk, chunk_cnt,
                           rows, count);
    });
    hipLaunchKernelGGL(k_val_rank, dim3(ceil_div(M, 256), N), dim3(256), 0, st, rows, count, M, top_k, sel);
    hipLaunchKernelGGL(k_val_gather, dim3(N), dim3(64), 0, st, rows, count, M, top_k, sel, out, out_count);
    return YOLO_LAUNCH_CHECK();
}

// pred: fp32 [N][top_k][6] (first 5 columns used), count: int32 [N]; gt: fp32 [total][5] (cx, cy, w, h, cls) with
// image i owning rows gt_off[i] .. gt_off[i+1]; counters: int64 [5 + 4 nc], accumulated (zero them to reset);
// skip_empty_gt: ignore images without targets (the reference's validation loop does, train_model.py:326-328);
// status: int32[1] set to 1 if an image has more than 1024 targets (counters then incomplete).
int yolo_val_match(const float* pred, const int* count, int N, int top_k, const float* gt, const int* gt_off, double iou_thr,
                   int nc, int skip_empty_gt, long long* counters, int* status, hipStream_t st) {
    if (N < 1 || top_k < 1 || nc < 1) return YOLO_ERR_ARG;
    hipLaunchKernelGGL(k_val_match, dim3(N), dim3(64), 0, st, pred, count, top_k, gt, gt_off, iou_thr, nc, skip_empty_gt,
                       counters, status);
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
