#!/usr/bin/env python3
"""Generate tests/golden/metrics.npz by RUNNING THE REFERENCE's DetectionMetrics (build container only).

Usage (from the repo root):  python tests/golden/gen_metrics_golden.py [/root/reference]

Imports src/training/metrics.py of the reference read-only; the .npz holds inputs (per-image prediction and
target rows, concatenated with offsets) and the reference's counters after feeding them through
DetectionMetrics.update (metrics.py:68-157) in image order, for two IoU thresholds.  Images cover: no
predictions, no targets, neither, class mismatches, class ids outside [0, num_classes), several predictions
competing for one target, identical targets (first index wins the tie), IoUs on both sides of the threshold.
"""
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from src.training.metrics import DetectionMetrics, box_iou_batch      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NC = 8
g = torch.Generator().manual_seed(77)


def rnd(*shape):
    return torch.rand(*shape, generator=g)


def image(n_gt, n_pred, jitter, wrong_cls=0.2, extra=2):
    """targets (n_gt,5) and predictions (n_pred+extra,5): jittered copies of random targets + random boxes"""
    gt = torch.cat([rnd(n_gt, 2) * 500 + 60, rnd(n_gt, 2) * 120 + 20, torch.floor(rnd(n_gt, 1) * NC)], 1)
    if n_gt == 0 or n_pred == 0:
        pred = torch.zeros(0, 5)
    else:
        src = torch.randint(0, n_gt, (n_pred,), generator=g)
        pred = gt[src].clone()
        pred[:, :2] += (rnd(n_pred, 2) - 0.5) * jitter * pred[:, 2:4]
        pred[:, 2:4] *= 1 + (rnd(n_pred, 2) - 0.5) * jitter
        flip = rnd(n_pred) < wrong_cls
        pred[flip, 4] = torch.floor(rnd(int(flip.sum())) * NC)
    if extra:
        stray = torch.cat([rnd(extra, 2) * 500 + 60, rnd(extra, 2) * 120 + 20, torch.floor(rnd(extra, 1) * NC)], 1)
        pred = torch.cat([pred, stray], 0)
        pred = pred[torch.randperm(pred.shape[0], generator=g)]
    return pred, gt


cases = []
for n_gt, n_pred, jit in [(5, 7, 0.3), (12, 30, 0.5), (1, 4, 0.2), (20, 100, 0.6), (3, 3, 0.05), (70, 100, 0.4), (9, 9, 0.8)]:
    cases.append(image(n_gt, n_pred, jit))
cases.append(image(0, 0, 0.0, extra=3))                  # predictions only
p, t = image(4, 0, 0.0, extra=0)                         # targets only
cases.append((p, t))
cases.append((torch.zeros(0, 5), torch.zeros(0, 5)))     # neither
# identical targets: the first one must be taken first; out-of-range and negative class ids
t = torch.tensor([[100., 100., 40., 40., 2.], [100., 100., 40., 40., 2.], [300., 300., 50., 80., 9.], [200., 120., 30., 30., -1.]])
p = torch.tensor([[101., 100., 40., 40., 2.], [100., 101., 40., 40., 2.], [100., 100., 40., 40., 2.], [300., 300., 50., 80., 9.],
                  [200., 120., 30., 30., -1.], [400., 400., 10., 10., 11.]])
cases.append((p, t))
# IoU exactly on the threshold side: boxes shifted so that IoU is just below / above 0.5
t = torch.tensor([[100., 100., 60., 60., 1.], [300., 100., 60., 60., 1.]])
p = torch.tensor([[120., 100., 60., 60., 1.], [319.9, 100., 60., 60., 1.], [100., 100., 60., 60., 3.]])
cases.append((p, t))

arrs = {"num_classes": np.asarray(NC)}
arrs["pred"] = torch.cat([c[0] for c in cases]).numpy()
arrs["gt"] = torch.cat([c[1] for c in cases]).numpy()
arrs["pred_off"] = np.cumsum([0] + [c[0].shape[0] for c in cases]).astype(np.int32)
arrs["gt_off"] = np.cumsum([0] + [c[1].shape[0] for c in cases]).astype(np.int32)
arrs["iou_first"] = box_iou_batch(cases[0][0][:, :4], cases[0][1][:, :4]).numpy()
for thr in (0.5, 0.45):
    m = DetectionMetrics(NC, iou_threshold=thr)
    per_image = []
    for pr, gt in cases:
        m.update(pr, gt)
        per_image.append([m.total_predictions, m.total_ground_truths, m.true_positives, m.false_positives, m.false_negatives])
    tag = f"thr{thr}"
    arrs[tag + ":scalars_after_each"] = np.asarray(per_image, dtype=np.int64)
    arrs[tag + ":class_tp"] = m.class_tp.numpy()
    arrs[tag + ":class_fp"] = m.class_fp.numpy()
    arrs[tag + ":class_fn"] = m.class_fn.numpy()
    arrs[tag + ":class_gt"] = m.class_gt_count.numpy()
    res = m.compute()
    arrs[tag + ":compute_keys"] = np.asarray(sorted(res))
    arrs[tag + ":compute_vals"] = np.asarray([res[k] for k in sorted(res)], dtype=np.float64)
    cm = m.get_class_metrics(2)
    arrs[tag + ":class2_keys"] = np.asarray(sorted(cm))
    arrs[tag + ":class2_vals"] = np.asarray([cm[k] for k in sorted(cm)], dtype=np.float64)
arrs["_torch_version"] = np.asarray(torch.__version__)
np.savez_compressed(os.path.join(OUT, "metrics.npz"), **arrs)
print("wrote metrics.npz", {k: v.shape for k, v in arrs.items() if hasattr(v, "shape")})
print(arrs["thr0.5:scalars_after_each"][-1], arrs["thr0.45:scalars_after_each"][-1])
