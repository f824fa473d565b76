"""Gradients of one train step with the lazily joined / grouped weight-gradient stream against the per-layer
fork/join form (same inputs, eager and graph-replayed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import functions as F_
from src.model.model_builder import Model
from src.model.losses import YoloDFLQFLoss
from src.training.graph_step import TrainStepRunner
from src.training.fused_adamw import HipAdamW

dev = "cuda"
torch.manual_seed(0)
model = Model(width=[3, 32, 64, 128, 256, 512], depth=[1] * 6, csp=[False, True], num_classes=80).to(dev).train()
class SmoothLoss:
    """a loss without discrete decisions (the detection loss's anchor assignment flips under 1e-6 perturbations)"""
    def __call__(self, preds, packed, anchors, strides):
        loss = (preds.float() * torch.linspace(0.5, 1.5, preds.shape[1], device=preds.device).view(1, -1, 1)).square().mean()
        class LD: _scalars = None
        return loss, LD()
crit = SmoothLoss()
opt = HipAdamW(model.parameters(), lr=0.0)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(8, 3, 640, 640, device=dev, generator=g)
gts = [torch.cat([torch.rand(5, 2, device=dev) * 600 + 20, torch.rand(5, 2, device=dev) * 100 + 20, torch.randint(0, 80, (5, 1), device=dev).float()], 1) for _ in range(8)]
from src.model.losses import PackedTargets
packed = PackedTargets(gts, dev)

def grads(lazy, group, graph):
    os.environ["YOLO_LAZY_JOIN"] = "1" if lazy else "0"
    F_.WGRAD_GROUP = group
    r = TrainStepRunner(model, crit, opt, precision="bfloat16", use_graph=graph)
    if graph:
        r.capture(img, packed, warmup=2)
        for _ in range(3):
            r.step()
    else:
        opt.zero_grad(set_to_none=True)
        r._fwd_bwd(img, packed)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

base = grads(False, 1, False)
for lazy, group, graph in [(False, 1, False), (True, 1, False), (True, 2, False), (True, 4, False), (True, 1, True), (True, 4, True), (False, 1, True)]:
    got = grads(lazy, group, graph)
    worst, wn = 0.0, ""
    gmax = max(float(v.abs().max()) for v in base.values())
    for n in base:      # tensors whose true gradient is zero (a shift in front of a BatchNorm) hold only rounding noise
        d = float((got[n] - base[n]).abs().max()) / (float(base[n].abs().max()) + 1e-3 * gmax)
        if d > worst:
            worst, wn = d, n
    print(f"lazy={lazy} group={group} graph={graph}: worst relative gradient difference {worst:.2e} ({wn})", flush=True)
