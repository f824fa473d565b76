"""The reference's image transforms (src/data/transforms.py:4-24) on the DEVICE, for a whole batch at once.

    get_train_transforms()  ToImage -> RandomHorizontalFlip(0.5) -> Resize((640, 640)) -> ColorJitter(0.2, 0.2, 0.2, 0.1)
                            -> ToDtype(float32, scale=True) -> Normalize(ImageNet)
    get_val_transforms()    ToImage -> Resize -> ToDtype -> Normalize

The reference runs them per image with torchvision on DataLoader workers (PIL decode + ~40 ms of CPU per image); here
the workers only decode, and `BatchTransform` uploads the decoded uint8 images of a batch (any sizes) and runs flip +
antialiased bilinear resize + colour jitter + normalisation as a handful of launches (csrc/image_prep.hip).  The random
decisions are drawn on the host in torchvision's order (flip: one `torch.rand(1)`; jitter: `torch.randperm(4)` then one
uniform per factor), boxes follow the geometry (XYWH: x' = W - x - w for a flip, then scaling by 640/W, 640/H) and the
batch leaves in the format of src/data/collate.py: (images[N,3,S,S], [target dicts with "boxes" (Mi, 5)])."""
import math

import numpy as np
import torch

from src.hipops import ops

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _uniform(lo, hi):
    return float(torch.empty(1).uniform_(lo, hi))


class BatchTransform:
    """Callable(images, targets) -> (batch on `device`, targets).  images: decoded RGB images as uint8 (H, W, 3) arrays /
    tensors or PIL images; targets: dicts with "boxes" (M, 4) XYWH pixels and "labels" (M, 1) (the reference dataset's
    items before its transform) or None."""

    def __init__(self, train, size=640, device="cuda", dtype=torch.float32, flip_p=0.5, brightness=0.2, contrast=0.2,
                 saturation=0.2, hue=0.1, mean=MEAN, std=STD):
        self.train, self.size, self.device, self.dtype = train, size, torch.device(device), dtype
        self.flip_p, self.jit = flip_p, (brightness, contrast, saturation, hue)
        self.mean, self.std = mean, std
        self._pinned = None                 # reused pinned upload buffer (one memcpy per image into it, one async H2D per batch)
        self._uploaded = None

    def sample(self):
        """One image's random decisions, drawn like torchvision draws them: (flip, order, factors)."""
        if not self.train:
            return False, (), (1.0, 1.0, 1.0, 0.0)
        flip = bool(torch.rand(1) < self.flip_p)
        order = tuple(int(i) for i in torch.randperm(4))
        b, c, s, h = self.jit
        fac = (_uniform(max(0.0, 1 - b), 1 + b), _uniform(max(0.0, 1 - c), 1 + c), _uniform(max(0.0, 1 - s), 1 + s), _uniform(-h, h))
        return flip, order, fac

    @staticmethod
    def _as_u8(img):
        if isinstance(img, torch.Tensor):
            t = img
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)))
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("BatchTransform takes decoded RGB images as uint8 (H, W, 3)")
        return t.contiguous()

    def __call__(self, images, targets=None, params=None):
        imgs = [self._as_u8(i) for i in images]
        params = params if params is not None else [self.sample() for _ in imgs]
        recs, off = [], 0
        for t, (flip, order, fac) in zip(imgs, params):
            h, w = int(t.shape[0]), int(t.shape[1])
            recs.append((off, h, w, flip, order, fac))
            off += h * w * 3
        if self.device.type == "cuda":
            if self._pinned is None or self._pinned.numel() < off:
                self._pinned = torch.empty(int(off * 1.25), dtype=torch.uint8).pin_memory()
            elif self._uploaded is not None:
                self._uploaded.synchronize()        # the previous batch's upload has left the buffer
            for t, r in zip(imgs, recs):
                self._pinned[r[0]:r[0] + t.numel()].copy_(t.reshape(-1))
            flat = self._pinned[:off].to(self.device, non_blocking=True)
            self._uploaded = torch.cuda.Event()
            self._uploaded.record(torch.cuda.current_stream(self.device))
        else:
            flat = torch.cat([t.reshape(-1) for t in imgs])
        batch = ops.image_prep(flat, recs, self.size, self.train and any(len(p[1]) for p in params), self.dtype, self.mean, self.std)
        out_t = None
        if targets is not None:
            out_t = []
            for tg, (_, h, w, flip, _, _) in zip(targets, recs):
                b = tg["boxes"].clone().float().reshape(-1, 4)
                if flip:
                    b[:, 0] = w - (b[:, 0] + b[:, 2])
                b[:, [0, 2]] *= self.size / w
                b[:, [1, 3]] *= self.size / h
                new = {k: v for k, v in tg.items() if k not in ("boxes", "labels")}
                new["boxes"] = torch.cat([b, tg["labels"].float().reshape(-1, 1)], 1)      # dataset_loader.py:76
                out_t.append(new)
        return batch, out_t


def get_train_transforms(size=640, device="cuda", dtype=torch.float32):
    return BatchTransform(True, size, device, dtype)


def get_val_transforms(size=640, device="cuda", dtype=torch.float32):
    """Batch form; `Model.inference` applies it to a single PIL image (reference :16-24)."""
    return BatchTransform(False, size, device, dtype)
