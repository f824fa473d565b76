"""Deterministic, regenerable parameter store for the oracle (test infrastructure).

The reference initialises with PyTorch's default RNG-driven init
(src/model/model_blocks.py:27-28, src/model/head.py:66-74).  Golden fixtures
cannot carry 10 MB of weights, so both the fixture generator and the tests
fill parameters with ``det_tensor``: a value stream that depends only on
(key, shape, kind, seed) and is therefore identical wherever it is rebuilt.
"""
import math
import zlib

import torch


def det_tensor(key: str, shape, kind: str, seed: int = 0) -> torch.Tensor:
    """kind: conv | bias | cls_bias | bn_weight | bn_bias | bn_mean | bn_var | dfl | counter"""
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    if kind == "conv":
        fan_in = max(1, int(torch.tensor(shape[1:]).prod()))
        return torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
    if kind == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    if kind == "cls_bias":  # src/model/head.py:68-74
        return torch.full(shape, math.log(0.01 / 0.99))
    if kind == "bn_weight":
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if kind == "bn_bias":
        return 0.1 * torch.randn(shape, generator=g)
    if kind == "bn_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if kind == "bn_var":
        return 1.0 + 0.2 * torch.rand(shape, generator=g)
    if kind == "dfl":  # src/model/model_blocks.py:273-275
        return torch.arange(shape[1], dtype=torch.float32).view(shape)
    if kind == "counter":
        return torch.zeros(shape, dtype=torch.long)
    raise ValueError(kind)


def kind_of(key: str) -> str:
    """Classify a reference state-dict key (names: SURVEY.md section 5)."""
    if key.endswith("num_batches_tracked"):
        return "counter"
    if key == "head.dfl.conv.weight":
        return "dfl"
    if ".norm." in key:
        return {"weight": "bn_weight", "bias": "bn_bias",
                "running_mean": "bn_mean", "running_var": "bn_var"}[key.rsplit(".", 1)[1]]
    if key.endswith(".bias"):
        return "cls_bias" if key.startswith("head.cls.") else "bias"
    return "conv"


def det_fill_(state_dict, seed: int = 0):
    """Overwrite every entry of a state dict (reference's or ours) in place."""
    with torch.no_grad():
        for k, v in state_dict.items():
            v.copy_(det_tensor(k, v.shape, kind_of(k), seed).to(v.dtype))
    return state_dict


class ParamStore(dict):
    """``{key: tensor}`` that materialises missing entries on first use."""

    def __init__(self, seed: int = 0, requires_grad: bool = False):
        super().__init__()
        self.seed = seed
        self.requires_grad = requires_grad

    def want(self, key: str, shape) -> torch.Tensor:
        if key not in self:
            t = det_tensor(key, shape, kind_of(key), self.seed)
            if self.requires_grad and t.is_floating_point() and ".running_" not in key \
                    and key != "head.dfl.conv.weight":
                t.requires_grad_(True)
            self[key] = t
        t = self[key]
        assert tuple(t.shape) == tuple(shape), (key, tuple(t.shape), tuple(shape))
        return t
