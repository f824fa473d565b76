"""Time the elementwise BN/activation leaves through the C-ABI (run under rocprofv3 --kernel-trace for kernel times)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops

def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

for (n, c, h, w) in [(32,128,80,80), (32,64,160,160), (32,128,160,160), (32,256,40,40), (32,512,20,20), (32,32,320,320)]:
    mk = lambda: torch.randn(n, c, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, dout = mk(), mk()
    f = lambda k: torch.randn(c, device="cuda") * 0.1 + k
    scale, shift, mean, invstd, gamma = f(1), f(0), f(0), f(1), f(1)
    mb = y.numel() * 2 / 1e6
    us_f = t(lambda: ops.bn_act_fwd(y, scale, shift, 1))
    us_b = t(lambda: ops.bn_act_bwd(dout, y, scale, shift, mean, invstd, gamma, 1))
    print(f"({n},{c},{h},{w}) {mb:6.1f} MB  fwd {us_f:7.1f} us {2*mb/us_f:6.2f} TB/s | bwd {us_b:7.1f} us {5*mb/us_b:6.2f} TB/s(5 passes)", flush=True)
