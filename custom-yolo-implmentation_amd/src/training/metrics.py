"""Validation-time detection counters (reference: src/training/metrics.py:44-191; outside the accelerated
path).  Same greedy rule -- predictions in order, each takes the unmatched same-class
target of highest IoU, TP when that IoU >= threshold -- but each image costs ONE device->host copy and a
numpy loop instead of an .item() per (prediction, target) pair."""
from typing import Dict

import numpy as np
import torch


def box_iou_batch(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    """(N,4) x (M,4) centre-xywh boxes -> (N,M) IoU with the reference's 1e-6 in the denominator."""
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

    def _bump(self, arr, classes):
        c = np.asarray(classes, dtype=np.int64)
        c = c[(c >= 0) & (c < self.num_classes)]
        np.add.at(arr, c, 1)

    def update(self, predictions: torch.Tensor, targets: torch.Tensor, pred_scores: torch.Tensor = None,
               score_threshold: float = 0.5):
        if predictions.numel() == 0 and targets.numel() == 0:
            return
        if pred_scores is not None and predictions.numel() > 0:
            predictions = predictions[pred_scores >= score_threshold]
        n, m = (predictions.shape[0] if predictions.numel() else 0), (targets.shape[0] if targets.numel() else 0)
        if n == 0:
            self.false_negatives += m
            cls = targets[:, 4].long().cpu().numpy()
            self._bump(self.class_fn, cls), self._bump(self.class_gt_count, cls)
            return
        if m == 0:
            self.false_positives += n
            self._bump(self.class_fp, predictions[:, 4].long().cpu().numpy())
            return
        iou = box_iou_batch(predictions[:, :4].float(), targets[:, :4].float()).cpu().numpy()
        pc, tc = predictions[:, 4].long().cpu().numpy(), targets[:, 4].long().cpu().numpy()
        taken = np.zeros(m, dtype=bool)
        for i in range(n):
            cand = np.where((tc == pc[i]) & ~taken & (iou[i] > 0), iou[i], -1.0)
            j = int(cand.argmax())
            if cand[j] >= self.iou_threshold and cand[j] > 0:
                taken[j] = True
                self.true_positives += 1
                self._bump(self.class_tp, [pc[i]])
            else:
                self.false_positives += 1
                self._bump(self.class_fp, [pc[i]])
        self.false_negatives += int((~taken).sum())
        self._bump(self.class_gt_count, tc)
        self._bump(self.class_fn, tc[~taken])
        self.total_predictions += n
        self.total_ground_truths += m

    def compute(self) -> Dict[str, float]:
        tp, fp, fn = self.true_positives, self.false_positives, self.false_negatives
        precision, recall = tp / (tp + fp + 1e-6), tp / (tp + fn + 1e-6)
        cp = self.class_tp / (self.class_tp + self.class_fp + 1e-6)
        valid = self.class_gt_count > 0
        return {"precision": float(precision), "recall": float(recall),
                "f1_score": float(2 * precision * recall / (precision + recall + 1e-6)),
                "mAP": float(cp[valid].mean()) if valid.any() else 0.0, "true_positives": int(tp),
                "false_positives": int(fp), "false_negatives": int(fn),
                "total_predictions": int(self.total_predictions), "total_ground_truths": int(self.total_ground_truths)}

    def get_class_metrics(self, class_id: int) -> Dict[str, float]:
        tp, fp, fn = self.class_tp[class_id], self.class_fp[class_id], self.class_fn[class_id]
        p, r = tp / (tp + fp + 1e-6), tp / (tp + fn + 1e-6)
        return {"precision": float(p), "recall": float(r), "f1_score": float(2 * p * r / (p + r + 1e-6)),
                "true_positives": int(tp), "false_positives": int(fp), "false_negatives": int(fn),
                "ground_truths": int(self.class_gt_count[class_id])}
