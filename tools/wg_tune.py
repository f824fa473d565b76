"""Tuning harness for the weight-gradient plan (tile, workgroup count): sustained timing per candidate."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import lib, ops

def timeit(fn, secs=0.25):
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
    (32, 32, 320, 320, 64, 3, 2), (32, 128, 160, 160, 128, 3, 2), (32, 256, 80, 80, 256, 3, 2), (32, 256, 40, 40, 512, 3, 2)]
which = sys.argv[1:] 
for (n, cin, h, w, cout, k, s) in shapes:
    x = torch.randn(n, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    oh, ow = ops.conv_out_hw(h, w, k, s)
    dy = torch.randn(n, cout, oh, ow, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    lib.call("yolo_wgrad_tune_set", 0, 0, 0, 0)
    base = timeit(lambda: ops.conv_wgrad(x, dy, k, s, torch.float32))
    best = []
    tiles = [(a, b) for a in (1, 2, 3, 4) for b in (1, 2, 3, 4)] if k == 1 else [(1, 1), (1, 2), (2, 1), (2, 2)]
    tiles = [(a, b) for a, b in tiles if 32 * (a - 1) < cout and 32 * (b - 1) < cin]
    for (to, ti) in tiles:
        for blocks in (64, 128, 256, 512, 1024, 2048):
            lib.call("yolo_wgrad_tune_set", to, ti, blocks, 2)
            try:
                us = timeit(lambda: ops.conv_wgrad(x, dy, k, s, torch.float32), 0.1)
            except Exception as e:
                continue
            best.append((us, to, ti, blocks))
    best.sort()
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    print(f"({n},{cin},{h},{w})->{cout} k{k}s{s} {mb:6.0f} MB ideal {mb/4.5:6.1f} us | default {base:7.1f} | " +
          "  ".join(f"{u:6.1f}@{a}x{b}/{bl}" for u, a, b, bl in best[:4]), flush=True)
