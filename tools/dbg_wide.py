"""Repeatability of one forced conv variant with the 16-byte epilogue stores on / off: the same launch REPS times, every
result compared bit for bit with the first one and element-wise with the fp32 reference.
Usage: dbg_wide.py [reps]   (shape and tuning of tests/test_gpu_conv_variants.py::test_gather_kernel...[96-200-1-1-0-32])"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
import torch.nn.functional as F
from src.hipops import lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
DEV, BF = "cuda", torch.bfloat16
CASES = [(3, 96, 200, 37, 41, 1, 1, 32, 0, 32), (3, 96, 200, 37, 41, 1, 1, 32, 1, 33), (3, 64, 128, 37, 41, 3, 1, 64, 0, 64),
         (3, 96, 200, 37, 41, 1, 1, 0, -1, 5)]


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


for (n, cin, cout, h, w, k, s, bn, dma, seed) in CASES:
    lib.call("yolo_conv_tune_set", bn, -1, 0, dma, 0, 0, 0, 0)
    x = rnd((n, cin, h, w), seed + 1)
    wt = rnd((cout, cin, k, k), seed + 2, (cin * k * k) ** -0.5).float()
    wp = ops.pack_weights(wt.to(DEV), k, s, 0, BF)
    xbuf = ops.new_nhwc(n, cin + 32, h, w, BF, DEV).fill_(3.0)
    xd = xbuf[:, 16:16 + cin]
    xd.copy_(x.to(DEV))
    oh, ow = ops.conv_out_hw(h, w, k, s)
    y_ref = F.conv2d(x.float(), wt.to(BF).float(), None, s, k // 2).to(DEV)
    lim = 2.0 ** -8 * y_ref.abs() + 1e-3 * float(y_ref.abs().max())
    for wide in (2, 0, 1):
        lib.call("yolo_conv_wide_set", wide)
        ybuf = ops.new_nhwc(n, cout + 16, oh, ow, BF, DEV).fill_(5.0)
        yv = ybuf[:, 8:8 + cout]
        first, nondet, off = None, 0, 0
        for r in range(reps):
            acc = ops.bn_acc_new(cout, DEV)
            y = ops.conv_fwd(xd, wp, None, cout, k, s, acc, out=yv)
            if first is None:
                first = y.clone()
            else:
                nondet += int((y != first).sum())
            off += int(((y.float() - y_ref).abs() > lim).sum())
            if r % 500 == 499:
                torch.cuda.synchronize()
        plan = lib.query("yolo_conv2d_plan", n, h, w, cin, oh, ow, cout, k, s, 0, 0, lib.BF16)
        print(f"case cin {cin} cout {cout} k {k} bn {bn} dma {dma} plan {plan} wide {wide}: {reps} launches, elements differing from "
              f"the first launch {nondet}, outside the reference band {off}", flush=True)
lib.call("yolo_conv_tune_set", 0, -1, 0, -1, -1, 0, 0, 0)
lib.call("yolo_conv_wide_set", 2)
