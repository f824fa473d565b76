"""Marginal cost per patch of k_wgrad2: the same layer with the workgroup count forced (slabs = blocks / tiles), both prefetch
depths.  time = fixed (launch, prologue, partial store, reduce) + patches per workgroup x cost per patch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from conv_layers import graph_time
from src.hipops import ops, lib
N = 32
shapes = [(128, 128, 20, 20, 3, 1), (256, 512, 40, 40, 3, 2), (512, 512, 20, 20, 1, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for cin, cout, h, w, k, s in shapes:
    oh, ow = ops.conv_out_hw(h, w, k, s)
    x = torch.randn(N, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, cout, oh, ow, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = torch.empty(cout, cin, k, k, device="cuda")
    npatch = N * ((oh + 3) // 4) * ((ow + 7) // 8)
    for pf in (1, 4):
        for blocks in (32, 64, 128, 256, 512, 1024):
            lib.call("yolo_wgrad_tune_set", 0, 0, blocks, 1)
            lib.call("yolo_wgrad_tune_pf", pf)
            plan = lib.query("yolo_conv2d_wgrad_plan", N, h, w, cin, oh, ow, cout, k, s, lib.BF16)
            nslab = plan % 100000
            us = graph_time(lambda: ops.conv_wgrad(x, dy, k, s, torch.float32, out=dw))
            print(f"{cin}->{cout} {h}x{w} k{k}s{s} pf{pf} blocks {blocks:5d} slabs {nslab:4d} patches/wg {npatch / nslab:6.1f}  {us:7.1f} us", flush=True)
    lib.call("yolo_wgrad_tune_set", 0, 0, 0, 0); lib.call("yolo_wgrad_tune_pf", 0)
