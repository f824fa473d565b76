"""Validation-time detection counters on the device (reference: src/training/metrics.py:44-191).

Same greedy rule as the reference -- predictions in order, each takes the unmatched same-class target of highest
IoU (strict >, so IoU 0 never matches and the first target wins a tie), TP when that IoU >= threshold -- but the
reference spends an `.item()` per (prediction, target) pair in Python loops; here one kernel launch handles a whole
batch (one wave per image, `yolo_val_match`) and the counters stay in device memory until `compute()` /
`get_class_metrics()` / `sync()` fetch them with one copy.  `update()` keeps the reference's per-image signature;
`update_batch()` takes the packed output of `decode_predictions_packed` (src/training/train_model.py).
There is no host implementation: the tensors must live on the GPU like every other op of this package.
"""
from typing import Dict, List

import numpy as np
import torch

from src.hipops import ops


def box_iou_batch(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    """(N,4) x (M,4) centre-xywh boxes -> (N,M) IoU with the reference's 1e-6 in the denominator (:6-41)."""
    def corners(b):
        return b[:, :2] - b[:, 2:4] / 2, b[:, :2] + b[:, 2:4] / 2
    (l1, r1), (l2, r2) = corners(boxes1), corners(boxes2)
    wh = (torch.min(r1[:, None], r2[None]) - torch.max(l1[:, None], l2[None])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    a1, a2 = (r1 - l1).prod(1), (r2 - l2).prod(1)
    return inter / (a1[:, None] + a2[None] - inter + 1e-6)


class DetectionMetrics:
    def __init__(self, num_classes: int, iou_threshold: float = 0.5):
        self.num_classes, self.iou_threshold = num_classes, iou_threshold
        self.reset()

    def reset(self):
        self.total_predictions = self.total_ground_truths = 0
        self.true_positives = self.false_positives = self.false_negatives = 0
        z = lambda: np.zeros(self.num_classes, dtype=np.int64)
        self.class_tp, self.class_fp, self.class_fn, self.class_gt_count = z(), z(), z(), z()
        self._dev = None            # (counters int64 [5 + 4 nc], status int32 [1]) on the device of the first update

    # ------------------------------------------------------------------------------------------ device side
    def _state(self, device):
        if self._dev is None:
            self._dev = (ops.zero_(torch.empty(5 + 4 * self.num_classes, dtype=torch.int64, device=device)),
                         ops.zero_(torch.empty(1, dtype=torch.int32, device=device)))
        elif self._dev[0].device != device:
            raise RuntimeError("DetectionMetrics: updates must stay on one device between reset() calls")
        return self._dev

    def update_batch(self, rows: torch.Tensor, count: torch.Tensor, targets: List[torch.Tensor],
                     skip_empty_targets: bool = True):
        """rows fp32 [N][K][6] (cx, cy, w, h, cls, score), count int32 [N] (valid rows per image), targets: N tensors
        (Mi, 5).  skip_empty_targets mirrors the reference's validation loop (train_model.py:326-328), which does not
        call update() for an image without ground truth."""
        dev = rows.device
        counters, status = self._state(dev)
        sizes = [int(t.shape[0]) if t.numel() else 0 for t in targets]
        if len(sizes) != rows.shape[0]:
            raise ValueError("one target tensor per image expected")
        parts = [t.reshape(-1, 5).to(device=dev, dtype=torch.float32) for t, s in zip(targets, sizes) if s]
        gt = torch.cat(parts) if parts else torch.zeros(1, 5, dtype=torch.float32, device=dev)
        off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).to(dev, non_blocking=True)
        ops.val_match(rows.contiguous(), count, gt.contiguous(), off, self.iou_threshold, self.num_classes,
                      skip_empty_targets, counters, status)

    def update(self, predictions: torch.Tensor, targets: torch.Tensor, pred_scores: torch.Tensor = None,
               score_threshold: float = 0.5):
        """Reference signature (:68-69): one image, predictions (N,5) / targets (M,5) rows [x, y, w, h, class]."""
        if predictions.numel() == 0 and targets.numel() == 0:
            return
        dev = predictions.device if predictions.numel() else targets.device
        if pred_scores is not None and predictions.numel() > 0:
            predictions = predictions[pred_scores >= score_threshold]
        n = predictions.shape[0] if predictions.numel() else 0
        rows = torch.zeros(1, max(n, 1), 6, dtype=torch.float32, device=dev)
        if n:
            rows[0, :, :5] = predictions[:, :5].to(device=dev, dtype=torch.float32)
        count = torch.full((1,), n, dtype=torch.int32, device=dev)
        self.update_batch(rows, count, [targets], skip_empty_targets=False)

    def sync(self):
        """Fetch the device counters (one copy) into the reference's attributes and clear the device side."""
        if self._dev is None:
            return self
        counters, status = self._dev
        host = counters.cpu().numpy().copy()
        if int(status.cpu()[0]) != 0:
            raise RuntimeError("DetectionMetrics: an image had more than 1024 targets (yolo_val_match limit)")
        ops.zero_(counters)
        nc = self.num_classes
        self.total_predictions += int(host[0])
        self.total_ground_truths += int(host[1])
        self.true_positives += int(host[2])
        self.false_positives += int(host[3])
        self.false_negatives += int(host[4])
        self.class_tp += host[5:5 + nc]
        self.class_fp += host[5 + nc:5 + 2 * nc]
        self.class_fn += host[5 + 2 * nc:5 + 3 * nc]
        self.class_gt_count += host[5 + 3 * nc:5 + 4 * nc]
        return self

    # ------------------------------------------------------------------------------------------ results (:159-205)
    def compute(self) -> Dict[str, float]:
        self.sync()
        tp, fp, fn = self.true_positives, self.false_positives, self.false_negatives
        precision, recall = tp / (tp + fp + 1e-6), tp / (tp + fn + 1e-6)
        # the reference keeps per-class counters as fp32 tensors: the class precision is an fp32 quotient
        ctp, cfp = self.class_tp.astype(np.float32), self.class_fp.astype(np.float32)
        cp = ctp / (ctp + cfp + np.float32(1e-6))
        valid = self.class_gt_count > 0
        return {"precision": float(precision), "recall": float(recall),
                "f1_score": float(2 * (precision * recall) / (precision + recall + 1e-6)),
                "mAP": float(cp[valid].mean(dtype=np.float32)) if valid.any() else 0.0, "true_positives": int(tp),
                "false_positives": int(fp), "false_negatives": int(fn),
                "total_predictions": int(self.total_predictions), "total_ground_truths": int(self.total_ground_truths)}

    def get_class_metrics(self, class_id: int) -> Dict[str, float]:
        self.sync()
        tp, fp, fn = (np.float32(a[class_id]) for a in (self.class_tp, self.class_fp, self.class_fn))
        eps = np.float32(1e-6)
        p, r = tp / (tp + fp + eps), tp / (tp + fn + eps)
        return {"precision": float(p), "recall": float(r), "f1_score": float(np.float32(2) * (p * r) / (p + r + eps)),
                "true_positives": int(tp), "false_positives": int(fp), "false_negatives": int(fn),
                "ground_truths": int(self.class_gt_count[class_id])}


def compute_average_iou(predictions: List[torch.Tensor], targets: List[torch.Tensor]) -> float:
    """Mean over predictions of their best IoU with any target of the image (:208-236)."""
    total, pairs = 0.0, 0
    for pred, target in zip(predictions, targets):
        if pred.numel() == 0 or target.numel() == 0:
            continue
        total += box_iou_batch(pred[:, :4].float(), target[:, :4].float()).max(dim=1)[0].sum().item()
        pairs += pred.size(0)
    return total / (pairs + 1e-6)
