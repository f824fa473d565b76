"""Row-block conv kernel (conv_rows.hip) against the default choice on the 20- / 40-pixel-wide 3x3 layers of the step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import time
import torch
from src.hipops import lib, ops

def tune(bn=0, tap_inner=-1, halo=-1, dma=-1, ring=0, bm=0, nst=0, bk=0):
    lib.call("yolo_conv_tune_set", bn, tap_inner, halo, dma, ring, bm, nst, bk)

sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_layers import graph_time as timeit

shapes = [(32, 64, 80, 80, 64), (32, 32, 80, 80, 64), (32, 64, 80, 80, 32), (32, 128, 80, 80, 64), (32, 128, 80, 80, 128), (32, 64, 160, 160, 64),
          (32, 64, 40, 40, 64), (32, 64, 40, 40, 128), (32, 128, 40, 40, 64), (32, 256, 40, 40, 64), (32, 128, 40, 40, 128),
          (32, 256, 40, 40, 256), (32, 64, 20, 20, 64), (32, 128, 20, 20, 128), (32, 512, 20, 20, 64), (32, 256, 20, 20, 256)]
for (n, cin, h, w, cout) in shapes:
    x = torch.randn(n, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weights(wt, 3, 1, 0, torch.bfloat16)
    wb = ops.pack_weights(wt, 3, 1, 1, torch.bfloat16)
    acc = ops.bn_acc_new(cout, "cuda")
    y = torch.empty_like(dy)
    for name, fn in (("fwd", lambda: ops.conv_fwd(x, wp, None, cout, 3, 1, acc, out=y)), ("dgrad", lambda: ops.conv_dgrad(dy, wb, cin, h, w, 3, 1))):
        res = []
        for hv in (0, 8, 12, -1, 14):
            tune(0, -1, hv)
            res.append(timeit(fn))
        tune()
        print(f"{name:5s} {cin:4d}->{cout:4d} {h}x{w}  gather ring {res[0]:6.1f} us  80px x 64 {res[1]:6.1f}  160px x 64 {res[2]:6.1f}  default {res[3]:6.1f}  10x16 px x 64 {res[4]:6.1f}", flush=True)
