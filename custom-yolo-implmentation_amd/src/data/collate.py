"""Batch format of the hot path (reference: src/data/collate.py): images stacked, targets kept as a list."""
import torch


def collate_fn(batch):
    return torch.stack([b[0] for b in batch]), [b[1] for b in batch]


def raw_collate_fn(batch):
    """Decoded images of different sizes cannot be stacked: lists of (uint8 HWC image, target); the device transform
    (src/data/transforms.py::BatchTransform) turns them into collate_fn's format."""
    return [b[0] for b in batch], [b[1] for b in batch]
