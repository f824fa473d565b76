"""Batch format of the hot path (reference: src/data/collate.py): images stacked, targets kept as a list."""
import torch


def collate_fn(batch):
    return torch.stack([b[0] for b in batch]), [b[1] for b in batch]
