// The three module-level helpers of the reference's loss file as stand-alone differentiable ops (fp32):
//   bbox_iou (src/model/losses.py:9-40, with its b1_y2 = h + cy/2 slip), quality_focal_loss (:46-57),
//   distribution_focal_loss (:63-78).
// The training step never calls them -- YoloDFLQFLoss is one fused pass (loss.hip) -- they exist so that notebook-level
// code written against the reference's names runs on the device.  One launch forward, one backward each.
#include "common.h"

namespace {

// d min(a,b)/da (torch.minimum / maximum split a tie half and half)
__device__ __forceinline__ float w_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// iou[m] from centre-xywh rows; with g != null also the gradients: db1 / db2 [M][4] = g[m] * d iou / d box
__global__ void k_bbox_iou(const float* __restrict__ b1, const float* __restrict__ b2, int M, float* __restrict__ iou,
                           const float* __restrict__ g, float* __restrict__ db1, float* __restrict__ db2) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float4 p = reinterpret_cast<const float4*>(b1)[m], q = reinterpret_cast<const float4*>(b2)[m];
    const float px1 = p.x - p.z / 2, py1 = p.y - p.w / 2, px2 = p.x + p.z / 2, py2 = p.w + p.y / 2;   // reference :20
    const float gx1 = q.x - q.z / 2, gy1 = q.y - q.w / 2, gx2 = q.x + q.z / 2, gy2 = q.y + q.w / 2;
    const float rw = fminf(px2, gx2) - fmaxf(px1, gx1), rh = fminf(py2, gy2) - fmaxf(py1, gy1);
    const float iw = fmaxf(rw, 0.f), ih = fmaxf(rh, 0.f);
    const float inter = iw * ih;
    const float w1 = px2 - px1, h1 = py2 - py1, w2 = gx2 - gx1, h2 = gy2 - gy1;
    const float U = w1 * h1 + w2 * h2 - inter + 1e-6f;
    if (iou != nullptr) iou[m] = inter / U;
    if (g == nullptr) return;
    const float go = g[m];
    const float d_inter = go * (1.f / U + inter / (U * U)), d_area = -go * inter / (U * U);
    const float d_iw = rw >= 0.f ? d_inter * ih : 0.f, d_ih = rh >= 0.f ? d_inter * iw : 0.f;      // clamp(min=0)
    // x direction: ix2 = min(px2, gx2), ix1 = max(px1, gx1), iw = ix2 - ix1
    float d_px2 = d_iw * w_lt(px2, gx2), d_gx2 = d_iw * w_lt(gx2, px2);
    float d_px1 = -d_iw * w_lt(gx1, px1), d_gx1 = -d_iw * w_lt(px1, gx1);
    float d_py2 = d_ih * w_lt(py2, gy2), d_gy2 = d_ih * w_lt(gy2, py2);
    float d_py1 = -d_ih * w_lt(gy1, py1), d_gy1 = -d_ih * w_lt(py1, gy1);
    d_px2 += d_area * h1; d_px1 -= d_area * h1; d_py2 += d_area * w1; d_py1 -= d_area * w1;
    d_gx2 += d_area * h2; d_gx1 -= d_area * h2; d_gy2 += d_area * w2; d_gy1 -= d_area * w2;
    // corners -> (cx, cy, w, h); box1's y2 = h + cy/2
    reinterpret_cast<float4*>(db1)[m] = make_float4(d_px1 + d_px2, d_py1 + 0.5f * d_py2, 0.5f * (d_px2 - d_px1), d_py2 - 0.5f * d_py1);
    reinterpret_cast<float4*>(db2)[m] = make_float4(d_gx1 + d_gx2, d_gy1 + d_gy2, 0.5f * (d_gx2 - d_gx1), 0.5f * (d_gy2 - d_gy1));
}

// forward: out[0] += sum of terms / M (out zeroed by the caller); backward (g != null): dx, dt = g * d loss / d (logit, target)
__global__ __launch_bounds__(256) void k_qfl(const float* __restrict__ x, const float* __restrict__ t, long n, int M, float beta,
                                              float* __restrict__ out, const float* __restrict__ g, float* __restrict__ dx,
                                              float* __restrict__ dt) {
    __shared__ float red[4];
    float acc = 0.f;
    const float inv = 1.f / (float)M;
    const float go = g != nullptr ? *g : 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float s = 1.f / (1.f + __expf(-x[i])), tt = t[i];
        const float ls = __logf(s + 1e-12f), l1 = __logf(1.f - s + 1e-12f);
        const float a = powf(1.f - s, beta), b = powf(s, beta);
        if (g == nullptr) {
            acc -= tt * a * ls + (1.f - tt) * b * l1;
        } else {
            const float a1 = beta * powf(1.f - s, beta - 1.f), b1 = beta * powf(s, beta - 1.f);
            const float dpos = tt * (-a1 * ls + a / (s + 1e-12f));
            const float dneg = (1.f - tt) * (b1 * l1 - b / (1.f - s + 1e-12f));
            dx[i] = -go * inv * (dpos + dneg) * s * (1.f - s);
            dt[i] = -go * inv * (a * ls - b * l1);
        }
    }
    if (g != nullptr) return;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * inv);
}

// one thread per row of C logits; forward: out[0] += row loss / M; backward: dx [M][C], dt [M]
__global__ void k_dfl(const float* __restrict__ x, const float* __restrict__ t, int M, int C, float* __restrict__ out,
                      const float* __restrict__ g, float* __restrict__ dx, float* __restrict__ dt) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.f;
    if (m < M) {
        const float* r = x + (long)m * C;
        const float tv = t[m];
        const int lo = (int)tv, hi = lo + 1;                  // .long(): truncation
        const float wl = (float)hi - tv, wr = tv - (float)lo;
        float mx = r[0];
        for (int j = 1; j < C; ++j) mx = fmaxf(mx, r[j]);
        float se = 0.f;
        for (int j = 0; j < C; ++j) se += __expf(r[j] - mx);
        const float lse = mx + __logf(se);
        const float xl = r[lo < C ? lo : C - 1], xh = r[hi < C ? hi : C - 1];
        if (g == nullptr) {
            term = (lse - xl) * wl + (lse - xh) * wr;
        } else {
            const float go = *g / (float)M;
            for (int j = 0; j < C; ++j)
                dx[(long)m * C + j] = go * ((wl + wr) * __expf(r[j] - lse) - (j == lo ? wl : 0.f) - (j == hi ? wr : 0.f));
            dt[m] = go * (xl - xh);
        }
    }
    if (g != nullptr) return;
    term = wave_sum(term);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, term / (float)M);
}

}  // namespace

extern "C" {

// fp32 tensors; iou [M]; g null = forward, else backward into db1 / db2 [M][4] (iou may then be null)
int yolo_bbox_iou(const float* box1, const float* box2, int M, float* iou, const float* g, float* db1, float* db2, hipStream_t st) {
    if (M <= 0) return YOLO_OK;
    hipLaunchKernelGGL(k_bbox_iou, dim3(ceil_div(M, 256)), dim3(256), 0, st, box1, box2, M, iou, g, db1, db2);
    return YOLO_LAUNCH_CHECK();
}

// pred / target fp32 [M][C]; forward: out[1] (zeroed here); backward (g = device scalar grad of the loss): dpred, dtarget
int yolo_quality_focal_loss(const float* pred, const float* target, int M, int C, float beta, float* out, const float* g,
                            float* dpred, float* dtarget, hipStream_t st) {
    if (M <= 0 || C <= 0) return YOLO_ERR_ARG;
    const long n = (long)M * C;
    if (g == nullptr) {
        int rc = yolo_zero_async(out, sizeof(float), st);
        if (rc) return rc;
    }
    long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(k_qfl, dim3((unsigned)b), dim3(256), 0, st, pred, target, n, M, beta, out, g, dpred, dtarget);
    return YOLO_LAUNCH_CHECK();
}

// pred_dist fp32 [M][C] logits, target_val fp32 [M] in [0, C-1); forward: out[1]; backward: dpred [M][C], dtarget [M]
int yolo_distribution_focal_loss(const float* pred_dist, const float* target_val, int M, int C, float* out, const float* g,
                                 float* dpred, float* dtarget, hipStream_t st) {
    if (M <= 0 || C <= 1) return YOLO_ERR_ARG;
    if (g == nullptr) {
        int rc = yolo_zero_async(out, sizeof(float), st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_dfl, dim3(ceil_div(M, 256)), dim3(256), 0, st, pred_dist, target_val, M, C, out, g, dpred, dtarget);
    return YOLO_LAUNCH_CHECK();
}

}  // extern "C"
