"""TEMP: time single conv layers through the C-ABI (standalone timing of single conv layers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops

def run(n, cin, h, w, cout, k, s, reps=int(os.environ.get("REPS", "20"))):
    x = torch.randn(n, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, k, k, device="cuda", dtype=torch.float32) * 0.05
    wp = ops.pack_weights(wt, k, s, 0, torch.bfloat16)
    for _ in range(3): y = ops.conv_fwd(x, wp, None, cout, k, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = ops.conv_fwd(x, wp, None, cout, k, s)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    oh, ow = y.shape[2], y.shape[3]
    fl = 2.0 * n * oh * ow * cout * cin * k * k
    by = x.numel() * 2 + y.numel() * 2
    print(f"({n},{cin},{h},{w})->{cout} k{k}s{s}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s {by/us/1e3:7.0f} GB/s", flush=True)

for shp in [(32,128,160,160,128,3,2), (32,64,160,160,64,3,1), (32,96,160,160,128,1,1), (32,256,80,80,256,3,2), (32,32,320,320,32,1,1)]:
    run(*shp)
