"""Decode + NMS restatement (test infrastructure).  Citations: reference src/utils/model_utils.py,
src/model/model_builder.py, src/training/train_model.py.

``greedy_nms`` restates torchvision.ops.nms (third-party, torchvision==0.24.1 per the reference's
environment.yml:29; source absent from the reference checkout and from this image => "parity
unpinned", pinned instead by known-answer cases in tests/test_nms_known.py).  Published semantics:
visit boxes by descending score; keep a box unless an already-kept box has IoU > thr with it
(strict), IoU = inter / (area_a + area_b - inter) with no eps; return kept indices in visit order.
"""
import numpy as np
import torch

from .blocks import dfl as _dfl

MAX_WH = 7680      # model_utils.py:210
MAX_NMS = 30000    # model_utils.py:211


def xywh2xyxy(x):
    """model_utils.py:166-171 (dw = w/2 computed once, then subtract/add)."""
    y = torch.empty_like(x)
    dw, dh = x[..., 2] / 2, x[..., 3] / 2
    y[..., 0], y[..., 1] = x[..., 0] - dw, x[..., 1] - dh
    y[..., 2], y[..., 3] = x[..., 0] + dw, x[..., 1] + dh
    return y


def stable_desc_order(scores: torch.Tensor) -> torch.Tensor:
    """Descending by score, ties broken by lower original index.  The reference's
    ``argsort(descending=True)`` (model_utils.py:259) is unstable; this is the build's
    definition of the order on ties."""
    return torch.sort(scores, descending=True, stable=True)[1]


def greedy_nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """Kept indices (int64) in descending-score order; arithmetic in the boxes' dtype -> fp32."""
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long)
    order = stable_desc_order(scores).numpy()
    b = boxes.float().numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = (x2 - x1) * (y2 - y1)
    dead = np.zeros(len(b), dtype=bool)
    keep = []
    for pos, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou > np.float32(thr)]] = True
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, labels=(), max_det=300, nc=0):
    """Per-image class-aware NMS -> list of (n, 6) [x1,y1,x2,y2,conf,cls].  model_utils.py:174-279.
    The wall-clock abort (:212,275-277) is not restated: it must never trip in parity runs."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    mi = 4 + nc
    cand = prediction[:, 4:mi].amax(1) > conf_thres                       # :206
    multi_label &= nc > 1
    out = [torch.zeros((0, 6))] * bs
    for xi in range(bs):
        x = prediction[xi].transpose(0, -1)[cand[xi]]                     # :222
        if labels and len(labels[xi]):                                    # :225-231
            # the reference builds (len, nc+nm+5)-wide rows here and its torch.cat with the
            # (n, nc+4) candidates raises; mirrored as the same error class, not "fixed".
            raise RuntimeError("apriori labels: column mismatch (reference model_utils.py:227-231)")
        if not x.shape[0]:
            continue
        box = xywh2xyxy(x[:, :4])
        cls = x[:, 4:mi]
        if multi_label:                                                   # :240-242
            i, j = (cls > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, 4 + j, None], j[:, None].float()), 1)
        else:                                                             # :244-245
            conf, j = cls.max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if classes is not None:                                           # :248-249
            x = x[(x[:, 5:6] == torch.tensor(classes)).any(1)]
        if not x.shape[0]:
            continue
        x = x[stable_desc_order(x[:, 4])[:MAX_NMS]]                       # :259
        off = x[:, 5:6] * (0 if agnostic else MAX_WH)                     # :262
        keep = greedy_nms(x[:, :4] + off, x[:, 4], iou_thres)[:max_det]   # :263-265
        out[xi] = x[keep]
    return out


def inference_decode(ps, preds, anchors, strides, nc):
    """preds (N,64+nc,M) -> (N,4+nc,M): DFL -> dist2bbox(xywh) -> *stride, raw class logits.
    model_builder.py:123-136, model_utils.py:120-129."""
    box, cls = preds.split((64, nc), 1)
    d = _dfl(ps, box)
    lt, rb = d.split(2, 1)
    a = anchors.unsqueeze(0)
    x1y1, x2y2 = a - lt, a + rb
    return torch.cat((torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides, cls), 1)


def decode_predictions(preds, anchors, strides, conf_threshold=0.25, top_k=100):
    """Validation decode -> list of (k,5) [cx,cy,w,h,cls].  train_model.py:33-142 (no NMS)."""
    n, _, m = preds.shape
    dist = preds[:, :64].view(n, 4, 16, m).permute(0, 3, 1, 2).softmax(3)
    ltrb = (dist * torch.arange(16, dtype=preds.dtype)).sum(3)
    a = anchors.transpose(0, 1).unsqueeze(0)
    s = strides.transpose(0, 1).unsqueeze(0)
    x1y1, x2y2 = a - ltrb[..., :2], a + ltrb[..., 2:]
    boxes = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], 2) * s
    res = []
    for b in range(n):
        sc, ci = preds[b, 64:].transpose(0, 1).sigmoid().max(1)
        m_ = sc >= conf_threshold
        bb, ss, cc = boxes[b][m_], sc[m_], ci[m_]
        if bb.numel() == 0:
            res.append(torch.zeros(0, 5))
            continue
        if ss.numel() > top_k:
            top = torch.topk(ss, top_k)[1]
            bb, cc = bb[top], cc[top]
        res.append(torch.cat([bb, cc.unsqueeze(1).float()], 1))
    return res


# ---- validation counters: reference src/training/metrics.py ------------------------------------------------
def box_iou_batch(b1, b2):
    """(N,4) x (M,4) centre-xywh -> (N,M) IoU, fp32, 1e-6 added to the union.  metrics.py:6-41, op by op."""
    def corners(b):
        return b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    ax1, ay1, ax2, ay2 = corners(b1)
    bx1, by1, bx2, by2 = corners(b2)
    w = (torch.min(ax2[:, None], bx2[None]) - torch.max(ax1[:, None], bx1[None])).clamp(min=0)
    h = (torch.min(ay2[:, None], by2[None]) - torch.max(ay1[:, None], by1[None])).clamp(min=0)
    inter = w * h
    a1, a2 = (ax2 - ax1) * (ay2 - ay1), (bx2 - bx1) * (by2 - by1)
    return inter / (a1[:, None] + a2[None] - inter + 1e-6)


class MetricCounters:
    """DetectionMetrics state + update as plain loops.  metrics.py:52-66 (reset), :68-157 (update)."""

    def __init__(self, num_classes, iou_threshold=0.5):
        self.nc, self.thr = num_classes, iou_threshold
        self.total_predictions = self.total_ground_truths = 0
        self.tp = self.fp = self.fn = 0
        self.class_tp, self.class_fp = np.zeros(num_classes, np.int64), np.zeros(num_classes, np.int64)
        self.class_fn, self.class_gt = np.zeros(num_classes, np.int64), np.zeros(num_classes, np.int64)

    def _ok(self, c):
        return 0 <= c < self.nc

    def update(self, predictions, targets):
        n = predictions.shape[0] if predictions.numel() else 0
        m = targets.shape[0] if targets.numel() else 0
        if n == 0 and m == 0:                                   # :80
            return
        pc = [int(v) for v in predictions[:, 4].long()] if n else []
        tc = [int(v) for v in targets[:, 4].long()] if m else []
        if n == 0:                                              # :91-98: totals are NOT advanced on this path
            self.fn += m
            for c in tc:
                if self._ok(c):
                    self.class_fn[c] += 1
                    self.class_gt[c] += 1
            return
        if m == 0:                                              # :100-106
            self.fp += n
            for c in pc:
                if self._ok(c):
                    self.class_fp[c] += 1
            return
        iou = box_iou_batch(predictions[:, :4].float(), targets[:, :4].float())
        taken = set()
        for i in range(n):                                      # :117-145, predictions in the given order
            best, bj = 0.0, -1
            for j in range(m):
                if j in taken:
                    continue
                if pc[i] == tc[j] and float(iou[i, j]) > best:  # strict: first index wins a tie, IoU 0 never matches
                    best, bj = float(iou[i, j]), j
            if best >= self.thr and bj >= 0:                    # python float (double) comparison
                self.tp += 1
                taken.add(bj)
                if self._ok(pc[i]):
                    self.class_tp[pc[i]] += 1
            else:
                self.fp += 1
                if self._ok(pc[i]):
                    self.class_fp[pc[i]] += 1
        self.fn += m - len(taken)                               # :148-156
        for j in range(m):
            if self._ok(tc[j]):
                self.class_gt[tc[j]] += 1
                if j not in taken:
                    self.class_fn[tc[j]] += 1
        self.total_predictions += n
        self.total_ground_truths += m

    def scalars(self):
        return [self.total_predictions, self.total_ground_truths, self.tp, self.fp, self.fn]

    def compute(self):                                          # :159-191
        p = self.tp / (self.tp + self.fp + 1e-6)
        r = self.tp / (self.tp + self.fn + 1e-6)
        # the reference keeps the per-class counters as fp32 tensors: precision is an fp32 quotient
        ctp, cfp = torch.from_numpy(self.class_tp).float(), torch.from_numpy(self.class_fp).float()
        cp = ctp / (ctp + cfp + 1e-6)
        valid = torch.from_numpy(self.class_gt > 0)
        return {"precision": float(p), "recall": float(r), "f1_score": float(2 * (p * r) / (p + r + 1e-6)),
                "mAP": cp[valid].mean().item() if valid.sum() > 0 else 0.0, "true_positives": int(self.tp),
                "false_positives": int(self.fp), "false_negatives": int(self.fn),
                "total_predictions": int(self.total_predictions), "total_ground_truths": int(self.total_ground_truths)}
