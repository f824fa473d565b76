"""Which layer's forward output first differs between two identical training-mode passes (fp32)?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.model.model_blocks import Conv
from src.model.model_builder import Model

NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
amp = os.environ.get("PROBE_AMP", "0") == "1"
g = torch.Generator().manual_seed(12)
img = torch.randn(2, 3, 160, 160, generator=g).cuda()
torch.manual_seed(0)
model = Model(**NANO, num_classes=80).cuda().train()
outs = {}


def hook(name):
    def f(m, i, o):
        outs.setdefault(name, []).append(o.detach().float().clone())
    return f


for n, m in model.named_modules():
    if type(m) is Conv:
        m.register_forward_hook(hook(n))
for r in range(4):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        p, _, _ = model(img)
    outs.setdefault("preds", []).append(p.float().clone())
torch.cuda.synchronize()
for n, v in outs.items():
    d = [float((v[i] - v[0]).abs().max() / v[0].abs().max().clamp_min(1e-20)) for i in range(1, 4)]
    mx = float(v[0].abs().max())
    if max(d) > 0:
        print(f"{n:40s} max|x| {mx:10.3e}  rel diffs vs run 0: {d[0]:.2e} {d[1]:.2e} {d[2]:.2e}  shape {tuple(v[0].shape)}")
print("done (modules without a line are bit-identical across runs)")
