"""Anchor grid, BN folding, box helpers and class-aware NMS with the reference's signatures
(reference: src/utils/model_utils.py).  The per-step pieces run on the HIP kernels
(yolo_nms, yolo_head_decode); the small tensor helpers (dist2bbox, xywh2xyxy, box_iou) are kept as
device-agnostic one-liners for API parity -- nothing on the training/inference hot path calls them."""
import torch
from torch import nn

from src.hipops import ops


def autopad(k, p=None, d=1):
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def _anchor_grid(shapes_hw, strides, dtype, device, offset=0.5):
    pts, sts = [], []
    for (h, w), s in zip(shapes_hw, strides):
        sx = torch.arange(w, device=device, dtype=dtype) + offset
        sy = torch.arange(h, device=device, dtype=dtype) + offset
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((gx, gy), -1).view(-1, 2))
        sts.append(torch.full((h * w, 1), s, dtype=dtype, device=device))
    return torch.cat(pts), torch.cat(sts)


def make_anchors(x, strides, offset: float = 0.5):
    """(M,2) cell centres (x fastest) and (M,1) strides in the dtype/device of x[0] (reference :18-70)."""
    assert x is not None
    return _anchor_grid([tuple(t.shape[-2:]) for t in x], [float(s) for s in strides], x[0].dtype, x[0].device, offset)


_ANCHOR_CACHE = {}


def make_anchors_cached(shapes_hw, strides, dtype, device):
    """The reference rebuilds the grid twice per forward (head.py:94,112); it only depends on the map
    sizes, so it is built once per (shapes, dtype, device) and returned already transposed:
    anchors (2, M), strides (1, M)."""
    key = (shapes_hw, strides, dtype, str(device))
    hit = _ANCHOR_CACHE.get(key)
    if hit is None:
        a, s = _anchor_grid(shapes_hw, strides, dtype, device)
        hit = (a.transpose(0, 1), s.transpose(0, 1))
        _ANCHOR_CACHE[key] = hit
    return hit


def fuse_conv(conv: nn.Conv2d, norm: nn.Module):
    """Fold BatchNorm into the preceding conv: W' = diag(g/sqrt(var+eps)) W, b' = beta - g*mean/sqrt(var+eps)
    (reference :72-118).  One-off parameter transformation."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    with torch.no_grad():
        s = norm.weight.float() / torch.sqrt(norm.running_var.float() + norm.eps)
        fused.weight.copy_(conv.weight.float() * s.view(-1, 1, 1, 1))
        b0 = torch.zeros_like(s) if conv.bias is None else conv.bias.float()
        fused.bias.copy_(b0 * s + norm.bias.float() - norm.running_mean.float() * s)
    return fused


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):
    lt, rb = torch.split(distance, 2, dim)
    x1y1, x2y2 = anchor_points - lt, anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


def box_iou(box1, box2, eps=1e-7):
    (a1, a2), (b1, b2) = box1.unsqueeze(1).chunk(2, 2), box2.unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


def xywh2xyxy(x):
    assert x.shape[-1] == 4, f"input shape last dimension expected 4 but input shape is {x.shape}"
    y = torch.empty_like(x)
    dw, dh = x[..., 2] / 2, x[..., 3] / 2
    y[..., 0], y[..., 1] = x[..., 0] - dw, x[..., 1] - dh
    y[..., 2], y[..., 3] = x[..., 0] + dw, x[..., 1] + dh
    return y


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, labels=(), max_det=300, nc=0):
    """Class-aware NMS, one (n, 6) fp32 tensor [x1,y1,x2,y2,conf,cls] per image (reference :174-279).

    Runs entirely in yolo_nms on the GPU; one host read of the per-image counts shapes the output list.
    Deviations, by design: score ties are ordered by lower candidate index (the reference's argsort is
    unstable); the reference's wall-clock abort (:212,275-277) is not reproduced (it would make results
    timing-dependent)."""
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if labels and any(len(lb) for lb in labels):
        raise RuntimeError("apriori labels: the reference's rows are nc+5 wide and its torch.cat raises (:227-231)")
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    if prediction.shape[1] != 4 + nc:
        raise ValueError("mask channels (nm > 0) are not part of this detector")
    multi_label = bool(multi_label) and nc > 1
    rows, counts, status = ops.nms(prediction, nc, conf_thres, iou_thres, classes, agnostic, multi_label, max_det)
    host = torch.cat((counts, status)).tolist()          # the one device->host sync
    if host[-1]:
        raise RuntimeError("yolo_nms: more candidates than the kernel capacity; raise conf_thres")
    return [rows[i, :host[i]] for i in range(bs)]
