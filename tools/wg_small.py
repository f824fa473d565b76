"""A few small-map weight gradients, replayed: run under rocprofv3 --kernel-trace --stats to split k_wgrad2 / k_wgrad_reduce."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops
N = 32
shapes = [(128, 128, 20, 20, 3, 1), (256, 512, 40, 40, 3, 2), (64, 64, 40, 40, 3, 1), (512, 512, 20, 20, 1, 1), (256, 256, 80, 80, 3, 2)]
for cin, cout, h, w, k, s in shapes:
    oh, ow = ops.conv_out_hw(h, w, k, s)
    x = torch.randn(N, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, cout, oh, ow, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = torch.empty(cout, cin, k, k, device="cuda")
    for _ in range(20):
        ops.conv_wgrad(x, dy, k, s, torch.float32, out=dw)
    torch.cuda.synchronize()
    print(cin, cout, h, w, k, s, "plan", ops.lib.query("yolo_conv2d_wgrad_plan", N, h, w, cin, oh, ow, cout, k, s, ops.lib.BF16))
