"""Tuning harness for the conv forward / data-gradient tile width and K order: sustained timing per candidate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import lib, ops

def tune(bn=0, tap_inner=-1, halo=-1, dma=-1, ring=0, bm=0, nst=0, bk=0):
    lib.call("yolo_conv_tune_set", bn, tap_inner, halo, dma, ring, bm, nst, bk)

def timeit(fn, secs=0.15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 20

shapes = [  # (n, cin, h, w, cout, k, s)
    (32, 32, 320, 320, 32, 1, 1), (32, 96, 160, 160, 128, 1, 1), (32, 64, 160, 160, 64, 1, 1), (32, 128, 80, 80, 128, 1, 1),
    (32, 512, 80, 80, 128, 1, 1), (32, 192, 80, 80, 256, 1, 1), (32, 384, 40, 40, 256, 1, 1), (32, 256, 40, 40, 256, 1, 1),
    (32, 768, 20, 20, 512, 1, 1), (32, 512, 20, 20, 512, 1, 1), (32, 256, 20, 20, 256, 1, 1), (32, 128, 40, 40, 128, 1, 1),
    (32, 64, 160, 160, 64, 3, 1), (32, 128, 80, 80, 128, 3, 1), (32, 64, 80, 80, 64, 3, 1), (32, 256, 40, 40, 256, 3, 1),
    (32, 128, 40, 40, 128, 3, 1), (32, 256, 20, 20, 256, 3, 1),
    (32, 32, 320, 320, 64, 3, 2), (32, 128, 160, 160, 128, 3, 2), (32, 256, 80, 80, 256, 3, 2), (32, 256, 40, 40, 512, 3, 2)]
for (n, cin, h, w, cout, k, s) in shapes:
    x = torch.randn(n, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    oh, ow = ops.conv_out_hw(h, w, k, s)
    dy = torch.randn(n, cout, oh, ow, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wp = ops.pack_weights(wt, k, s, 0, torch.bfloat16)
    wb = ops.pack_weights(wt, k, s, 1, torch.bfloat16)
    for name, fn in (("fwd", lambda: ops.conv_fwd(x, wp, None, cout, k, s)), ("dgrad", lambda: ops.conv_dgrad(dy, wb, cin, h, w, k, s))):
        tune()
        base = timeit(fn)
        res = []
        for bn in (128, 64, 32):
            for ti in ((0, 1) if k == 3 else (0,)):
                tune(bn, ti, 0)
                res.append((timeit(fn, 0.08), bn, ti))
        for bn in (128, 64, 32):
            tune(bn, 0, 0, 1)
            res.append((timeit(fn, 0.08), f"{bn}dma", 0))
        if k == 3 and s == 1:
            for hv in (1, 2, 3, 4):
                tune(0, 0, hv)
                res.append((timeit(fn, 0.08), "halo", hv))
        res.sort()
        mb = (x.numel() + dy.numel()) * 2 / 1e6
        print(f"{name:5s} ({n},{cin},{h},{w})->{cout} k{k}s{s} {mb:5.0f} MB ideal {mb/4.5:6.1f} us | default {base:7.1f} | " +
              "  ".join(f"{u:6.1f}@bn{b}/ti{t}" for u, b, t in res[:5]), flush=True)
