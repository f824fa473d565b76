"""Depthwise 3x3 kernels (forward, data gradient, weight gradient) on the head / PSA shapes of preset s at 32 images:
time per launch against the one-read-one-write HBM time.   python tools/dw_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops

def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

tot = [0, 0, 0, 0]
for (c, h) in [(128, 80), (128, 80), (256, 40), (128, 40), (512, 20), (128, 20), (128, 20)]:
    x = torch.randn(32, c, h, h, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    w9 = torch.randn(c, 9, device="cuda") * 0.2
    ideal = 2 * x.numel() * 2 / 5e6
    a, b, cc = t(lambda: ops.dw_fwd(x, w9)), t(lambda: ops.dw_dgrad(dy, w9)), t(lambda: ops.dw_wgrad(x, dy))
    tot[0] += a; tot[1] += b; tot[2] += cc; tot[3] += ideal
    print(f"C {c:4d} {h}x{h}: fwd {a:6.1f}  dgrad {b:6.1f}  wgrad {cc:6.1f} us   (ideal at 5 TB/s {ideal:5.1f})", flush=True)
print(f"TOTAL fwd {tot[0]:.0f} dgrad {tot[1]:.0f} wgrad {tot[2]:.0f} ideal {tot[3]:.0f} us each")
