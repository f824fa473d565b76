"""Run-to-run reproducibility of one bf16 forward/backward of the nano model under different kernel / stream settings:
prints, per setting, the worst per-tensor relative L2 difference between repeated runs on the same input (smooth objective)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import functions as F_
from src.hipops import lib
from src.model.model_builder import Model

NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
res = int(os.environ.get("PROBE_RES", "160"))
g = torch.Generator().manual_seed(12)
img = torch.randn(2, 3, res, res, generator=g).cuda()
torch.manual_seed(0)
model = Model(**NANO, num_classes=80).cuda().train()
M = (res // 8) ** 2 + (res // 16) ** 2 + (res // 32) ** 2
ct = (torch.randn(2, 144, M, generator=g) / M ** 0.5).cuda()
names = [n for n, p in model.named_parameters() if p.requires_grad]


def run(amp=True):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        preds, a, s = model(img)
        loss = (preds.float() * ct).sum()
    loss.backward()
    torch.cuda.synchronize()
    return preds.detach().float().clone(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}


def compare(tag, amp=True, reps=4):
    p0, g0 = run(amp)
    worst, wname, pdiff = 0.0, "", 0.0
    nmax = max(float(v.norm()) for v in g0.values())
    for _ in range(reps):
        p1, g1 = run(amp)
        pdiff = max(pdiff, float((p1 - p0).abs().max() / p0.abs().max()))
        for n in g0:
            r = float((g1[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-2 * nmax))
            if r > worst:
                worst, wname = r, n
    print(f"{tag:40s} preds max-rel {pdiff:.3e}   worst grad rel-L2 {worst:.3e}  ({wname})", flush=True)


def tune(ring=-1):
    lib.call("yolo_conv_tune_set", 0, -1, -1, -1, ring, 0, 0, 0)


compare("fp32")
compare("bf16 default")
tune(0)
compare("bf16 ring off")
tune(1)
compare("bf16 ring forced everywhere")
tune(-1)
F_.OVERLAP_WGRAD = False
compare("bf16 no wgrad side stream")
F_.OVERLAP_WGRAD = True
F_.HEAD_TWO_STREAMS = False
compare("bf16 head on one stream")
F_.OVERLAP_WGRAD = False
compare("bf16 single stream everywhere")
tune(0)
compare("bf16 single stream, ring off")
os.environ["YOLO_HIP_CONV_ALGO"] = "1"
