"""fp32 restatement of YoloDFLQFLoss (test infrastructure).

Follows src/model/losses.py of the reference, including its quirks
(SURVEY.md "four facts"): plain-IoU soft target with the y2 = h + cy/2 slip
(:20), nearest *predicted-centre* assignment (:214-215), duplicate index ->
the later GT wins (:261), IoU not detached (:256-261), lambda_box unused (:275).
Differentiable through torch autograd so d loss / d preds is the backward oracle.
"""
import torch
import torch.nn.functional as F

REG_MAX = 16


def quirk_iou(p, g, eps=1e-6):
    """IoU of centre-xywh rows; box1's y2 is h + cy/2.  src/model/losses.py:17-40."""
    px1, py1 = p[:, 0] - p[:, 2] / 2, p[:, 1] - p[:, 3] / 2
    px2, py2 = p[:, 0] + p[:, 2] / 2, p[:, 3] + p[:, 1] / 2   # reference :20
    gx1, gy1 = g[:, 0] - g[:, 2] / 2, g[:, 1] - g[:, 3] / 2
    gx2, gy2 = g[:, 0] + g[:, 2] / 2, g[:, 1] + g[:, 3] / 2
    iw = (torch.min(px2, gx2) - torch.max(px1, gx1)).clamp(min=0)
    ih = (torch.min(py2, gy2) - torch.max(py1, gy1)).clamp(min=0)
    inter = iw * ih
    union = (px2 - px1) * (py2 - py1) + (gx2 - gx1) * (gy2 - gy1) - inter
    return inter / (union + eps)


def qfl_sum(logits, target, beta=2.0):
    """-(t (1-s)^b log(s+1e-12) + (1-t) s^b log(1-s+1e-12)).sum() / rows.  losses.py:51-57."""
    s = logits.sigmoid()
    pos = target * (1 - s).pow(beta) * torch.log(s + 1e-12)
    neg = (1 - target) * s.pow(beta) * torch.log(1 - s + 1e-12)
    return -(pos + neg).sum() / logits.size(0)


def dfl_side(logits16, t):
    """Two-bin cross entropy around a continuous target; mean over rows.  losses.py:69-78."""
    lo = t.long()
    hi = lo + 1
    ce_lo = F.cross_entropy(logits16, lo, reduction="none")
    ce_hi = F.cross_entropy(logits16, hi, reduction="none")
    return (ce_lo * (hi.float() - t) + ce_hi * (t - lo.float())).mean()


def decode_boxes(dist, anchors_a2, strides_a1):
    """(N,A,4,16) logits -> centre-xywh pixels (N,A,4).  losses.py:156-188."""
    bins = torch.arange(REG_MAX, dtype=dist.dtype)
    ltrb = (dist.softmax(3) * bins).sum(3)
    ax, ay, st = anchors_a2[None, :, 0], anchors_a2[None, :, 1], strides_a1[None, :, 0]
    x1, y1 = (ax - ltrb[..., 0]) * st, (ay - ltrb[..., 1]) * st
    x2, y2 = (ax + ltrb[..., 2]) * st, (ay + ltrb[..., 3]) * st
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], 2)


def assign(gt_xy, pred_xy):
    """Index of the nearest predicted centre for every GT (first minimum).  losses.py:214-215."""
    return torch.cdist(gt_xy, pred_xy).argmin(dim=1)


def dfl_qfl_loss(preds, gt_list, anchors, strides, num_classes, lambda_cls=1.0, lambda_dfl=1.5):
    """Returns (total, mean_dfl, mean_cls) as 0-d tensors.  losses.py:140-281.

    preds (N, 64+nc, A) any float dtype; gt_list: N tensors (Mi,5) [cx,cy,w,h,cls] pixels;
    anchors (2,A), strides (1,A).
    """
    n = preds.shape[0]
    p = preds.float().transpose(1, 2)                        # :142
    anc = anchors.transpose(0, 1).float()
    st = strides.transpose(0, 1).float()
    dist = p[:, :, :4 * REG_MAX].reshape(n, -1, 4, REG_MAX)
    scores = p[:, :, 4 * REG_MAX:]
    boxes = decode_boxes(dist, anc, st)
    tot_dfl = torch.zeros(())
    tot_cls = torch.zeros(())
    for b in range(n):
        tgt = torch.zeros_like(scores[b])
        gt = gt_list[b]
        if gt.numel() > 0:
            g = gt[:, :4].float()
            idx = assign(g[:, :2], boxes[b][:, :2])
            m_anc, m_st = anc[idx], st[idx][:, 0]
            x1, y1 = (g[:, 0] - g[:, 2] / 2) / m_st, (g[:, 1] - g[:, 3] / 2) / m_st
            x2, y2 = (g[:, 0] + g[:, 2] / 2) / m_st, (g[:, 1] + g[:, 3] / 2) / m_st
            t = torch.stack([m_anc[:, 0] - x1, m_anc[:, 1] - y1, x2 - m_anc[:, 0], y2 - m_anc[:, 1]], 1)
            t = t.clamp(0, REG_MAX - 1 - 0.01)                # :246
            md = dist[b][idx]
            tot_dfl = tot_dfl + sum(dfl_side(md[:, s], t[:, s]) for s in range(4)) / 4.0
            iou = quirk_iou(boxes[b][idx], g)
            rows = torch.zeros(g.shape[0], num_classes)
            rows = rows.scatter(1, gt[:, 4].long().unsqueeze(1), iou.unsqueeze(1))   # :259-260
            # :261 -- CPU index_put_ is sequential: for duplicate idx the later GT row wins the
            # forward value, while autograd still hands EVERY row grad_out[idx[row]] (so a GT that
            # lost the slot keeps a gradient into its IoU).  Same op => same semantics.
            tgt = tgt.index_put((idx,), rows)
        tot_cls = tot_cls + qfl_sum(scores[b], tgt)
    mean_dfl, mean_cls = tot_dfl / n, tot_cls / n            # :271-272 (every image counts)
    return lambda_dfl * mean_dfl + lambda_cls * mean_cls, mean_dfl, mean_cls
