/* libyolo_hip.so -- C ABI of the MI355X (gfx950) YOLO training hot path.
 *
 * The reference (DarylFernandes99/custom-yolo-implmentation) has no FFI: its hot path is stock
 * PyTorch ops called from src/model/*.py.  Each entry point below replaces the ATen/torchvision op
 * the reference reaches at the cited file:line; the host mirror under
 * custom-yolo-implmentation_amd/src/ binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (sole exception: yolo_nms' `classes`, a host array of <=32 ints
 *     copied into the kernel arguments); nothing here allocates, frees or synchronises
 *   - activations are NHWC: element (pixel p, channel c) at base[p*ld + c], ld >= C (channel slices of
 *     concat buffers are passed as base+offset with the buffer's ld); dtype codes below; parameters,
 *     statistics and reductions are fp32
 *   - return 0 on success, a hipError_t value or YOLO_ERR_* otherwise; never aborts
 *   - re-entrant; the caller's stream is explicit (autograd's backward thread and DDP hooks call in
 *     concurrently with the main thread).  Process-wide state is limited to (a) once-initialised, read-only
 *     caches (per-device kernel attributes, tile plans) and (b) TEST / TUNING overrides that no product
 *     code path sets: yolo_conv_tune_set / yolo_wgrad_tune_set (forced tile variants in
 *     tests/test_gpu_conv_variants.py, tools/) and the environment switches read once at first use
 *     (YOLO_CONV_TUNE, YOLO_WG_TUNE, YOLO_WG_BLOCKS, YOLO_WG_SCALE, YOLO_WG_FLOOR, YOLO_WG_XCD,
 *     YOLO_DGRAD2_PATCH, YOLO_ATTN_FUSED: A/B measurement only).  With none of them set, results depend
 *     on the arguments alone
 *
 * One prototype per line, `int|long|size_t name(args);` -- the Python loader parses this file.
 */
#ifndef YOLO_HIP_H
#define YOLO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define YOLO_F32 0
#define YOLO_BF16 1
#define YOLO_F16 2
#define YOLO_ERR_ARG 1001
#define YOLO_ERR_DTYPE 1002
#define YOLO_ACT_IDENTITY 0
#define YOLO_ACT_SILU 1

/* hash of this header's prototypes at build time; the loader refuses a library built against another header */
long yolo_abi_hash(void);

/* ---- layout / copies (replace .view/.transpose/torch.cat/.chunk copies: head.py:87,119, model_blocks.py:92,123-125,156,249-252, neck.py:41-44) */
int yolo_memset0(void* p, size_t bytes, hipStream_t st);
int yolo_ncm_to_nhwc(const void* src, int src_dtype, long sn, long sc, long off, void* dst, int dst_dtype, int ld, int N, int C, int HW, hipStream_t st);
int yolo_nhwc_to_ncm(const void* src, int src_dtype, int ld, void* dst, int dst_dtype, long sn, long sc, long off, int N, int C, int HW, hipStream_t st);
/* HeadPack in one launch each way: up to 8 NHWC branch tensors <-> the (N, no, M) prediction tensor (head.py:87,119) */
int yolo_head_group(int pack, int n, void* const* branches, const int* lds, const int* Cs, const int* HWs, const int* c_offs, const int* m_offs, void* preds, int cp, int M, int N, int dtype, hipStream_t st);
int yolo_copy_channels(const void* src, int ld_src, void* dst, int ld_dst, long npix, int C, int accumulate, int dtype, hipStream_t st);
int yolo_scale_inplace(void* x, long n, int dtype, const float* scale_dev, hipStream_t st);
/* gradient fan-in: dst = sum of 2..4 channel-slice tensors in one pass (autograd's accumulation where a tensor has several consumers: model_blocks.py:62,92,156,223-224, neck.py:41-44, head.py:87) */
int yolo_add_n(const void* s0, int ld0, const void* s1, int ld1, const void* s2, int ld2, const void* s3, int ld3, int nsrc, void* dst, int ld_dst, long npix, int C, int dtype, hipStream_t st);

/* ---- convolution (nn.Conv2d: model_blocks.py:27, head.py:50,60; its autograd dgrad/wgrad) */
int yolo_conv_kpad(int O, int I, int k, int stride, int mode, int cls);
long yolo_conv_dgrad_wbuf_elems(int O, int I, int k, int stride);
int yolo_conv_pack_weights(const void* w_oihw, int w_dtype, int O, int I, int k, int stride, int mode, void* out, int out_dtype, hipStream_t st);
int yolo_conv_unpack_wgrad(const float* dwp, int O, int I, int k, void* dw_oihw, int dw_dtype, hipStream_t st);
int yolo_pack_job_bytes();
int yolo_pack_job_count(int stride, int mode);
long yolo_pack_job_fill(void* jobs_host, const void* w, int w_dtype, void* out, int out_elem_bytes, int O, int I, int k, int stride, int mode, long start);
long yolo_pack_jobs_finalize(void* jobs_host, int njobs);
int yolo_pack_batched(const void* jobs_dev, int njobs, long nchunks, int out_dtype, hipStream_t st);
/* ---- optimizer: torch.optim.AdamW over all parameters in one launch; replaces torch's multi-tensor
   AdamW (reference: src/training/utils_train.py:34 get_optimizer, src/training/train_model.py:247-253 step).
   jobs: device copy of a host table of yolo_adamw_job_bytes() records; hyper = [lr, beta1, beta2, eps, weight_decay] (doubles)
   and step (one float, incremented here) live in device memory; grad_scale / found_inf: GradScaler protocol or null. */
int yolo_adamw_job_bytes();
int yolo_adamw_job_fill(void* jobs_host, int index, void* p, int p_dtype, const void* g, int g_dtype, float* m, float* v, long n);
long yolo_adamw_jobs_finalize(void* jobs_host, int njobs);
int yolo_adamw_jobs_set_grads(void* jobs_host, int njobs, const void* const* grads);
int yolo_adamw_step(const void* jobs_dev, int njobs, long nchunks, const double* hyper, float* step, const float* grad_scale, const float* found_inf, hipStream_t st);
/* fp16 dynamic loss scaling without the host (GradScaler's scale / step / update, train_model.py:195-208,247-253): amp_state = fp32 [scale, found_inf (0 on entry), last_found_inf]; the gradients carry the factor `scale` (see yolo_loss_dfl_qfl's grad_scale) */
int yolo_adamw_amp_step(const void* jobs_dev, int njobs, long nchunks, const double* hyper, float* step, float* amp_state, int* growth_tracker, float growth_factor, float backoff_factor, int growth_interval, hipStream_t st);
/* ---- gradient exchange: pack / unpack every parameter gradient into / out of the flat communication buffers in one launch
   (DistributedDataParallel's bucket copies: src/training/utils_train.py:190); jobs: device copy of a host table of
   yolo_copy_job_bytes() records, dst = src * scale with a dtype cast */
int yolo_copy_job_bytes();
int yolo_copy_job_fill(void* jobs_host, int index, const void* src, int src_dtype, void* dst, int dst_dtype, long n);
long yolo_copy_jobs_finalize(void* jobs_host, int njobs);
int yolo_multi_copy(const void* jobs_dev, int njobs, long nchunks, float scale, hipStream_t st);
int yolo_conv2d_fwd(const void* x, int ldx, const void* wp, const float* bias, void* y, int ldy, float* stats_acc, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, int algo, hipStream_t st);
/* inference (Model.fuse(), model_blocks.py:36-37): y = act(conv(x) + bias) (+ res) in one launch; returns 1 = no MFMA kernel for this shape, nothing launched */
int yolo_conv2d_fwd_act(const void* x, int ldx, const void* wp, const float* bias, const void* res, int ldr, void* y, int ldy, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int act, int dtype, hipStream_t st);
int yolo_conv2d_dgrad(const void* dy, int lddy, const void* wb, void* dx, int lddx, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int accumulate, int dtype, int algo, hipStream_t st);
/* dx = dgrad + dx + acc2 in one epilogue (a three-way gradient fan-in, stride 1): see model_blocks.py C3K2 */
int yolo_conv2d_dgrad_acc2(const void* dy, int lddy, const void* wb, void* dx, int lddx, const void* acc2, int ld2, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, int algo, hipStream_t st);
long yolo_conv2d_wgrad_ws_elems(const void* x, int ldx, const void* dy, int ldy, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, int algo);
int yolo_conv2d_wgrad(const void* x, int ldx, const void* dy, int ldy, float* ws, void* dw_oihw, int dw_dtype, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype, int algo, hipStream_t st);
/* kernel-selection overrides for tuning runs (tools/conv_tune.py, tools/wg_tune.py) and the variant-forcing parity tests
   (tests/test_gpu_conv_variants.py); process-wide, 0 / -1 = automatic; the training path never calls them.
   conv: bn in {0, 32, 64, 128}, tap_inner / dma in {-1, 0, 1}, halo in {-1, 0 (gather kernel), 1..4}, ring in {-1, 0 (off), 1},
   bm in {0, 64, 128}, nst in {0, 2, 3, 4} and bk in {0, 32, 64} (ring kernel's pixel tile, ring depth and K-step).  No reference counterpart (model_blocks.py:27). */
int yolo_conv_tune_set(int bn, int tap_inner, int halo, int dma, int ring, int bm, int nst, int bk);
int yolo_wgrad_tune_set(int to, int ti, int blocks, int min_per);
int yolo_wgrad_tune_pf(int pf);
int yolo_conv_wide_set(int on);   /* conv epilogue stores: 2 = 16-byte, lane-pair exchange by v_permlane16_swap (default); 1 = by ds_bpermute; 0 = 8-byte */
/* which kernel a forward (mode 0) / data-gradient (mode 1, parity class cls for stride 2) launch of this shape takes:
   kind * 1000 + width, kind 1 = gather MFMA kernel (width = channel tile 32/64/128), 2 = halo MFMA kernel (width = variant 1..4),
   3 = pipelined ring kernel (width = channel tile, + 500 for 64-pixel tiles), 0 = VALU kernels */
int yolo_conv2d_plan(int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int mode, int cls, int dtype);
/* weight-gradient plan of a shape: to * 1000000 + ti * 100000 + nslab (MFMA path), 0 = other paths */
long yolo_conv2d_wgrad_plan(int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int dtype);
/* stem (3 -> C, 3x3 stride 2: backbone.py:38): NCHW image unfolded to K = 27(+5) columns, then the 1x1 MFMA path */
int yolo_stem_im2col(const void* img, int img_dtype, void* col, int col_dtype, int N, int H, int W, int OH, int OW, hipStream_t st);
int yolo_stem_pack_weights(const void* w, int w_dtype, int Cout, void* out, int out_dtype, hipStream_t st);
int yolo_stem_unpack_wgrad(const float* dw32, int Cout, void* dw, int dw_dtype, hipStream_t st);
/* fused stem forward: conv straight from the NCHW fp32 image (no column tensor), BatchNorm statistics in the epilogue;
   bias (fp32 [Cout], optional) + act (0 / 1 = SiLU): the folded-BatchNorm inference form (model_blocks.py:36-37), without stats */
int yolo_stem_conv_eligible(int img_dtype, int dtype, int Cout);
int yolo_stem_conv_fwd(const float* img, const void* wp, void* y, int ldy, float* stats, int N, int H, int W, int OH, int OW, int Cout, const float* bias, int act, int dtype, hipStream_t st);
/* stem weight gradient straight from the image (autograd's conv2d weight gradient of backbone.py:38; no column tensor):
   partial = fp32 scratch [yolo_stem_wgrad_slabs()][Cout][32]; dw = OIHW (Cout,3,3,3) of dw_dtype */
int yolo_stem_wgrad_slabs();
int yolo_stem_wgrad(const float* img, const void* dy, int ldy, float* partial, void* dw, int dw_dtype, int N, int H, int W, int OH, int OW, int Cout, int dtype, hipStream_t st);
/* depthwise 3x3 (groups == channels: model_blocks.py:183, head.py:56,58) */
int yolo_dwconv3x3_fwd(const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C, int dtype, hipStream_t st);
int yolo_dwconv3x3_fwd_stats(const void* x, int ldx, const float* w, void* y, int ldy, float* stats, int N, int H, int W, int C, int dtype, hipStream_t st);
/* fused inference: y = act(dwconv(x) + bias), one launch (model_blocks.py:36-37 on a depthwise Conv); 1 = not taken, nothing launched */
int yolo_dwconv3x3_fwd_act(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int H, int W, int C, int act, int dtype, hipStream_t st);
int yolo_dwconv3x3_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int H, int W, int C, int accumulate, int dtype, hipStream_t st);
int yolo_dw_wgrad_nslab(int N, int H);
int yolo_dwconv3x3_wgrad(const void* x, int ldx, const void* dy, int ldy, float* dw, float* partial, int N, int H, int W, int C, int dtype, hipStream_t st);

/* ---- BatchNorm2d(eps 1e-3, momentum 0.03) + SiLU/Identity + residual add (model_blocks.py:28-34,62,223-224)
 * pdtype = element type of gamma / beta (and of dgamma / dbeta), bdtype = element type of the running statistics:
 * YOLO_F32 under DDP / autocast; under FSDP mixed precision the reference casts parameters and BatchNorm buffers to
 * the low-precision dtype (src/training/utils_train.py:84-89,146-153) and they are read / written as they are. */
int yolo_reduce_nblk(long npix, int C);
int yolo_bn_stats(const void* y, int ldy, long npix, int C, int dtype, float* partial, int nblk, hipStream_t st);
int yolo_bn_finalize(const float* partial, int nblk, long count, int C, const void* gamma, const void* beta, void* running_mean, void* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, int pdtype, int bdtype, hipStream_t st);
int yolo_bn_eval_coeffs(const void* gamma, const void* beta, const void* running_mean, const void* running_var, float eps, int C, float* scale, float* shift, int pdtype, int bdtype, hipStream_t st);
int yolo_sum_finalize(const float* partial, int nblk, int C, float* out, hipStream_t st);
int yolo_bn_act_fwd(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldres, void* out, int ldout, long npix, int C, int act, int dtype, hipStream_t st);
int yolo_bn_act_bwd_reduce(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift, const float* mean, const float* invstd, long npix, int C, int act, int dtype, float* partial, int nblk, hipStream_t st);
int yolo_bn_bwd_finalize(const float* partial, int nblk, long count, int C, const void* gamma, const float* mean, const float* invstd, void* dgamma, void* dbeta, float* coef, int pdtype, hipStream_t st);
int yolo_bn_act_bwd_apply(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift, const float* mean, const float* invstd, const float* coef, void* dy, int lddy, long npix, int C, int act, int dtype, hipStream_t st);
/* accumulator form used by the training path: sums live in a caller-zeroed fp32 acc[8][2][C]; no finalize launches */
int yolo_bn_acc_elems(int C);
int yolo_bn_stats_acc(const void* y, int ldy, long npix, int C, int dtype, float* acc, hipStream_t st);
int yolo_bn_finalize_acc(const float* acc, long count, int C, const void* gamma, const void* beta, void* running_mean, void* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, int pdtype, int bdtype, hipStream_t st);
int yolo_bn_act_fwd_train(const void* y, int ldy, const float* acc, long count, const void* gamma, const void* beta, void* running_mean, void* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, const void* res, int ldres, void* out, int ldout, long npix, int C, int act, int dtype, int pdtype, int bdtype, hipStream_t st);
int yolo_bn_bwd_reduce_acc(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift, long npix, int C, int act, int dtype, float* acc, hipStream_t st);
int yolo_bn_act_bwd_apply_train(const void* dout, int ldd, const void* y, int ldy, const float* scale, const float* shift, const void* gamma, const float* mean, const float* invstd, const float* acc, long count, void* dgamma, void* dbeta, void* dy, int lddy, long npix, int C, int act, int dtype, int pdtype, hipStream_t st);

/* ---- SPPF max pool (model_blocks.py:150-156) and nearest x2 upsample (neck.py:31,41-42) */
int yolo_maxpool5_fwd(const void* x, int ldx, void* out, int ldo, uint8_t* idx, int N, int H, int W, int C, int dtype, hipStream_t st);
int yolo_maxpool5_bwd(const void* dout, int ldd, const uint8_t* idx, void* dx, int ldx, int N, int H, int W, int C, int accumulate, int dtype, hipStream_t st);
int yolo_upsample2x_fwd(const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, int dtype, hipStream_t st);
int yolo_upsample2x_bwd(const void* dout, int ldd, void* dx, int ldx, int N, int H, int W, int C, int accumulate, int dtype, hipStream_t st);

/* ---- PSA attention core (model_blocks.py:186-197) */
size_t yolo_attn_stash_bytes(int N, int T_, int heads, int dtype);
size_t yolo_attn_workspace_bytes(int N, int T_, int heads, int dtype);
/* the same sizes for a given head shape (the fused dk=32 / dh=64 kernels keep only the row log-sum-exp) */
size_t yolo_attn_stash_bytes_for(int N, int T, int heads, int dk, int dh, int dtype);
size_t yolo_attn_workspace_bytes_for(int N, int T, int heads, int dk, int dh, int dtype);
int yolo_attn_fwd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, void* stash, void* ws, int N, int T_, int heads, int dk, int dh, float scale, int dtype, hipStream_t st);
/* forward with no backward to follow (inference / no-grad): no stash, any sequence length; 1 = not taken, nothing launched */
int yolo_attn_fwd_nograd(const void* qkv, int ldq, void* o, int ldo, void* vp, int ldv, int N, int T_, int heads, int dk, int dh, float scale, int dtype, hipStream_t st);
int yolo_attn_bwd(const void* qkv, int ldq, const void* o, int ldo, const void* d_o, int lddo, const void* d_vp, int lddv, const void* stash, void* ws, void* dqkv, int lddq, int N, int T_, int heads, int dk, int dh, float scale, int dtype, hipStream_t st);

/* ---- YoloDFLQFLoss forward+gradient (losses.py:93-281) */
size_t yolo_loss_workspace_bytes(int N, int A, int G);
int yolo_loss_dfl_qfl(const void* preds, const void* anchors, const void* strides, int dtype, int N, int nc, int A, const float* gt, const int* gt_off, const int* gt_img, int G, float lambda_dfl, float lambda_cls, void* dpreds, float* out, void* workspace, const float* grad_scale, hipStream_t st);

/* the reference's module-level loss helpers as stand-alone differentiable fp32 ops (notebook API; the training step uses
   the fused pass above): bbox_iou (losses.py:9-40, incl. the b1_y2 slip), quality_focal_loss (:46-57),
   distribution_focal_loss (:63-78).  g == null: forward; g = device gradient (per row for bbox_iou, one scalar for the
   two losses): backward */
int yolo_bbox_iou(const float* box1, const float* box2, int M, float* iou, const float* g, float* db1, float* db2, hipStream_t st);
int yolo_quality_focal_loss(const float* pred, const float* target, int M, int C, float beta, float* out, const float* g, float* dpred, float* dtarget, hipStream_t st);
int yolo_distribution_focal_loss(const float* pred_dist, const float* target_val, int M, int C, float* out, const float* g, float* dpred, float* dtarget, hipStream_t st);

/* ---- inference decode (model_builder.py:123-136) and class-aware NMS (model_utils.py:174-279, torchvision.ops.nms) */
int yolo_head_decode(const void* preds, const void* anchors, const void* strides, void* y, int N, int nc, int A, int dtype, hipStream_t st);
int yolo_dfl_expect(const void* x, void* y, int B, int A, int dtype, hipStream_t st);
int yolo_nms_capacity(int M, int nc, int multi_label);
size_t yolo_nms_workspace_bytes(int bs, int M, int nc, int multi_label);
int yolo_nms(const void* y, int dtype, int bs, int nc, int M, float conf_thres, float iou_thres, const int* classes, int n_classes, int agnostic, int multi_label, int max_det, float* out, int* out_count, int* status, void* workspace, hipStream_t st);

/* ---- validation on device: decode_predictions' per-image select (train_model.py:14-142: sigmoid best class,
   `>= conf`, top-k by score; rows = cx, cy, w, h, cls, score) on yolo_head_decode's output, and DetectionMetrics.update
   (metrics.py:68-157: greedy same-class IoU matching in prediction order) into int64 counters
   [total_pred, total_gt, tp, fp, fn, class_tp[nc], class_fp[nc], class_fn[nc], class_gt[nc]] */
size_t yolo_val_workspace_bytes(int N, int M, int top_k);
int yolo_val_select(const void* y, int dtype, int N, int nc, int M, float conf, int top_k, float* out, int* out_count, void* workspace, hipStream_t st);
int yolo_val_match(const float* pred, const int* count, int N, int top_k, const float* gt, const int* gt_off, double iou_thr, int nc, int skip_empty_gt, long long* counters, int* status, hipStream_t st);

/* ---- on-device input pipeline: the reference's training / validation transform (src/data/transforms.py:4-24:
   horizontal flip, Resize((S,S)) bilinear + antialias, ColorJitter, ToDtype(scale), Normalize) for a batch of decoded uint8 HWC
   images of different sizes in one launch sequence; random decisions are drawn on the host (table records) */
int yolo_prep_image_bytes();
int yolo_prep_image_fill(void* table_host, int index, long off, int H, int W, int flip, int o0, int o1, int o2, int o3, float brightness, float contrast, float saturation, float hue);
int yolo_image_prep(const void* src, const void* table_dev, int N, int S, int jitter, void* stage, float* means, void* out, int out_dtype, float m0, float m1, float m2, float s0, float s1, float s2, hipStream_t st);

/* ---- hardware self-tests (instruction semantics the tiled kernels assume; tests/test_gpu_selftest.py; no reference counterpart: model_blocks.py:1) */
int yolo_selftest_tr16(const void* tile_in, void* out, hipStream_t st);
int yolo_selftest_glds(const void* in128x16, void* out64x16, hipStream_t st);
/* do cross-lane exchanges of one workgroup (spam 1 ds_bpermute, 2 v_permlane16_swap, 3 ds traffic, 0 VALU) disturb another workgroup's LDS-DMA on the same CU? */
int yolo_selftest_dma_vs_xlane(const void* pattern4k, int blocks, int iters, int spam, void* errors, void* sink, hipStream_t st);
/* do LDS-DMA transfers retire in issue order on vmcnt when a younger one has nothing to fetch (kind 1 zero-size descriptor, 2 out-of-range offset, 0 real)? */
int yolo_selftest_dma_order(const void* src, long src_bytes, int blocks, int kind, void* errors2, hipStream_t st);
/* the three-slot LDS-DMA ring (counted vmcnt, raw barrier) in even workgroups beside cross-lane traffic in odd ones: mismatching dwords */
int yolo_selftest_ring_vs_xlane(const void* src, int steps, int blocks, int spam, void* errors, void* sink, hipStream_t st);
int yolo_selftest_permlane16(const void* in64, void* out128, hipStream_t st);
/* `rounds` device-wide barriers among `blocks` resident workgroups (bounded spin; *timed_out set if one gave up): the per-layer price of a one-launch conv + BatchNorm */
int yolo_selftest_grid_barrier(void* counters, int blocks, int rounds, int tree, int* timed_out, hipStream_t st);

#ifdef __cplusplus
}
#endif
#endif
