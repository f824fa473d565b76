"""Per-layer timing of the training-path BatchNorm kernels (forward normalize+act; backward reduce + apply) on the output
shapes of a preset's dense convs (graph-replayed, warm clocks) against bytes / 5 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from conv_layers import graph_time
from layer_shapes import conv_calls
from src.hipops import ops

N = 32
shapes = {}
for (kind, cin, cout, h, w, k, s), cnt in conv_calls(sys.argv[1] if len(sys.argv) > 1 else "s").items():
    if kind == "fwd":
        oh, ow = ops.conv_out_hw(h, w, k, s)
        shapes[(cout, oh, ow)] = shapes.get((cout, oh, ow), 0) + cnt
tot = [0.0, 0.0, 0.0, 0.0]
for (c, h, w), cnt in sorted(shapes.items(), key=lambda t: (-t[0][1], t[0][0])):
    mk = lambda: torch.randn(N, c, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, dout, out = mk(), mk(), mk()
    f = lambda k: torch.randn(c, device="cuda") * 0.1 + k
    gamma, beta, rm, rv = f(1), f(0), f(0), f(1).abs()
    acc_f = ops.bn_acc_new(c, "cuda")
    ops.bn_stats_acc(y, acc_f)
    o, mean, invstd, scale, shift = ops.bn_act_fwd_train(y, acc_f, gamma, beta, rm, rv, 0.03, 1e-3, 1, None, out)
    acc_b = ops.bn_acc_new(c, "cuda")
    us_f = graph_time(lambda: ops.bn_act_fwd_train(y, acc_f, gamma, beta, rm, rv, 0.03, 1e-3, 1, None, out))

    def bwd():
        ops.zero_(acc_b)
        ops.bn_act_bwd_train(dout, y, scale, shift, mean, invstd, gamma, 1, acc_b)
    us_b = graph_time(bwd)
    mb = y.numel() * 2 / 1e6
    idf, idb = 2 * mb / 5.0, 5 * mb / 5.0
    tot[0] += us_f * cnt; tot[1] += idf * cnt; tot[2] += us_b * cnt; tot[3] += idb * cnt
    print(f"x{cnt} C{c:4d} {h:3d}x{w:3d} {mb:6.1f} MB | fwd {us_f:7.1f} us ({2 * mb / us_f:5.2f} TB/s, x{us_f / idf:4.2f}) | "
          f"bwd(zero+acc+apply) {us_b:7.1f} us ({5 * mb / us_b:5.2f} TB/s, x{us_b / idb:4.2f})", flush=True)
print(f"TOTAL fwd {tot[0] / 1e3:.3f} ms (ideal {tot[1] / 1e3:.3f}), bwd {tot[2] / 1e3:.3f} ms (ideal {tot[3] / 1e3:.3f})")
