"""YoloDFLQFLoss with the reference's constructor and call signature (src/model/losses.py:84-93,277-281),
computed by one fused HIP pass (yolo_loss_dfl_qfl: value + d/dpreds) instead of ~25 elementwise ATen
kernels plus a Python loop of ~100 tiny ops per image.  Reference semantics kept on purpose:
plain-IoU soft target with the b1_y2 = h + cy/2 slip (:20), nearest predicted-centre assignment via the
cdist formula (:214-215), last-GT-wins on duplicate anchors (:261) with the IoU gradient still reaching
every GT, lambda_box accepted and unused (:275), DFL targets clamped to [0, 14.99] (:246)."""
import torch
import torch.nn as nn

from src.hipops import functions as F_


class _Helper(torch.autograd.Function):
    """forward / backward of one stand-alone helper: ops leaf `name`(inputs..., extra..., g=None | grad)."""

    @staticmethod
    def forward(ctx, name, extra, *tensors):
        from src.hipops import ops
        ctx.name, ctx.extra = name, extra
        ts = [t.detach().float().contiguous() for t in tensors]
        ctx.save_for_backward(*ts)
        return getattr(ops, name)(*ts, *extra)

    @staticmethod
    def backward(ctx, g):
        from src.hipops import ops
        grads = getattr(ops, ctx.name)(*ctx.saved_tensors, *ctx.extra, g=g.detach().float().contiguous())
        return (None, None) + tuple(grads)


def bbox_iou(box1, box2):
    """IoU of matching rows of two (M, 4) centre-xywh tensors -> (M,), differentiable.  Reference signature and
    arithmetic (src/model/losses.py:9-40), including its b1_y2 = h + cy/2 slip and the 1e-6 in the denominator."""
    return _Helper.apply("bbox_iou", (), box1, box2)


def quality_focal_loss(pred_scores, target_scores, beta=2.0):
    """-(t (1-s)^b log(s+1e-12) + (1-t) s^b log(1-s+1e-12)).sum() / M for (M, C) logits / soft targets
    (src/model/losses.py:46-57)."""
    return _Helper.apply("qfl", (float(beta),), pred_scores, target_scores)


def distribution_focal_loss(pred_dist, target_val):
    """Two-bin cross entropy around a continuous target, mean over the M rows of (M, reg_max) logits
    (src/model/losses.py:63-78)."""
    return _Helper.apply("dfl_loss", (), pred_dist, target_val)


class PackedTargets:
    """GT boxes of a batch flattened for the kernel: gt (G,5) fp32, offsets (N+1) int32, image id (G) int32."""

    def __init__(self, gt_boxes_list, device):
        counts = [int(g.shape[0]) if g.numel() > 0 else 0 for g in gt_boxes_list]
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        self.n_gt = offs[-1]
        self.n_img = len(counts)
        live = [g.reshape(-1, 5).to(device=device, dtype=torch.float32) for g, c in zip(gt_boxes_list, counts) if c]
        self.gt = torch.cat(live).contiguous() if live else torch.zeros((1, 5), dtype=torch.float32, device=device)
        img = [i for i, c in enumerate(counts) for _ in range(c)] or [0]
        self.gt_off = torch.tensor(offs, dtype=torch.int32).to(device, non_blocking=True)
        self.gt_img = torch.tensor(img, dtype=torch.int32).to(device, non_blocking=True)

    def as_tuple(self):
        return self.gt, self.gt_off, self.gt_img, self.n_gt


class StaticTargets:
    """PackedTargets with fixed-capacity device buffers that `load()` refills in place: the form a captured training
    step needs (the kernels read the live count from gt_off[N] on the device; rows past it are ignored)."""

    def __init__(self, n_img, capacity, device):
        self.n_img, self.capacity = n_img, capacity
        self.gt = torch.zeros((capacity, 5), dtype=torch.float32, device=device)
        self.gt_off = torch.zeros(n_img + 1, dtype=torch.int32, device=device)
        self.gt_img = torch.zeros(capacity, dtype=torch.int32, device=device)
        self._h_gt = torch.zeros((capacity, 5), dtype=torch.float32).pin_memory()
        self._h_off = torch.zeros(n_img + 1, dtype=torch.int32).pin_memory()
        self._h_img = torch.zeros(capacity, dtype=torch.int32).pin_memory()
        self.n_gt, self._uploaded = 0, None

    def fits(self, gt_boxes_list):
        """Host-side check only: would load() accept this batch?"""
        return len(gt_boxes_list) == self.n_img and \
            sum(int(g.shape[0]) if g.numel() > 0 else 0 for g in gt_boxes_list) <= self.capacity

    def load(self, gt_boxes_list):
        """Refill from a batch's list of (Mi, 5) tensors; False (nothing changed) if they do not fit."""
        if not self.fits(gt_boxes_list):
            return False
        counts = [int(g.shape[0]) if g.numel() > 0 else 0 for g in gt_boxes_list]
        total = sum(counts)
        if self._uploaded is not None:
            self._uploaded.synchronize()                # the previous upload has left the pinned buffers
        off = 0
        self._h_off[0] = 0
        for i, (g, c) in enumerate(zip(gt_boxes_list, counts)):
            if c:
                self._h_gt[off:off + c].copy_(g.reshape(-1, 5).to(dtype=torch.float32, device="cpu"))
                self._h_img[off:off + c] = i
            off += c
            self._h_off[i + 1] = off
        self.gt.copy_(self._h_gt, non_blocking=True)
        self.gt_off.copy_(self._h_off, non_blocking=True)
        self.gt_img.copy_(self._h_img, non_blocking=True)
        self._uploaded = torch.cuda.Event()
        self._uploaded.record(torch.cuda.current_stream(self.gt.device))
        self.n_gt = total
        return True

    def as_tuple(self):
        return self.gt, self.gt_off, self.gt_img, self.capacity


class LazyLossDict(dict):
    """{"total_loss","box_loss","cls_loss": float}.  The three scalars stay on the device until first
    read and then arrive with ONE device->host copy (the reference does three .item() syncs, :278-280)."""

    _KEYS = ("total_loss", "box_loss", "cls_loss")

    def __init__(self, scalars):
        super().__init__()
        self._scalars = scalars

    def _fill(self):
        if self._scalars is not None:
            vals = self._scalars.tolist()
            self._scalars = None
            for k, v in zip(self._KEYS, vals):
                dict.__setitem__(self, k, v)

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        return 3

    def __contains__(self, k):
        return k in self._KEYS

    def keys(self):
        self._fill()
        return dict.keys(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)


class YoloDFLQFLoss(nn.Module):
    def __init__(self, num_classes=171, lambda_box=1.5, lambda_cls=1.0, lambda_dfl=1.5, reg_max=16):
        super().__init__()
        if reg_max != 16:
            raise ValueError("the loss kernel is built for reg_max = 16 (the head's DFL width)")
        self.num_classes = num_classes
        self.lambda_box = lambda_box      # accepted, unused -- as in the reference
        self.lambda_cls = lambda_cls
        self.lambda_dfl = lambda_dfl
        self.reg_max = reg_max

    def forward(self, preds, gt_boxes_list, anchors, strides):
        """preds (N, 64+nc, M); gt_boxes_list: N tensors (Mi,5) [cx,cy,w,h,cls] in pixels, or a
        PackedTargets; anchors (2,M); strides (1,M) -> (loss 0-d tensor with grad, dict of 3 floats)."""
        packed = gt_boxes_list if isinstance(gt_boxes_list, (PackedTargets, StaticTargets)) \
            else PackedTargets(gt_boxes_list, preds.device)
        if packed.n_img != preds.shape[0]:
            raise ValueError("one GT tensor per image is required")
        total, scalars = F_.DflQflLoss.apply(preds, anchors, strides, packed.as_tuple(), self.num_classes,
                                             float(self.lambda_dfl), float(self.lambda_cls))
        return total, LazyLossDict(scalars)
