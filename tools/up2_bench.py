"""Stride-2 data gradient: patch kernel (conv_up2.hip) against the gather ring on the step's shapes.  Run once per setting:
YOLO_DGRAD2_PATCH=0 (ring) / 1 (patch wherever eligible) / unset (default choice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from src.hipops import ops
from conv_layers import graph_time

shapes = [(32, 32, 320, 320, 64), (32, 128, 160, 160, 128), (32, 128, 80, 80, 128), (32, 256, 80, 80, 256), (32, 256, 40, 40, 256),
          (32, 256, 40, 40, 512)]
if os.environ.get("UP2_ONLY"):
    shapes = [shapes[int(os.environ["UP2_ONLY"])]]
for (n, cin, h, w, cout) in shapes:
    dy = torch.randn(n, cout, h // 2, w // 2, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wb = ops.pack_weights(wt, 3, 2, 1, torch.bfloat16)
    t = graph_time(lambda: ops.conv_dgrad(dy, wb, cin, h, w, 3, 2))
    mb = (dy.numel() + n * cin * h * w) * 2 / 1e6
    print(f"dgrad {cin:4d}->{cout:4d} {h}x{w} k3s2  {t:7.1f} us   ideal {mb / 5:6.1f} us", flush=True)
