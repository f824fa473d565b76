"""Per-layer timing of the conv kernels on the shapes of a preset's training step (graph-replayed launches, warm clocks),
against each layer's own roofline: max(FLOPs / 2.5 PF, bytes / 5 TB/s).  Usage: conv_layers.py [preset] [fwd|dgrad|wgrad ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from layer_shapes import conv_calls
from src.hipops import ops

N = int(os.environ.get("LAYERS_N", "32"))
REP = 20


def graph_time(fn):
    fn(); fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    return best


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "s"
    kinds = set(sys.argv[2:]) or {"fwd", "dgrad", "wgrad"}
    calls = conv_calls(preset)
    tot = {}
    for (kind, cin, cout, h, w, k, s), cnt in sorted(calls.items(), key=lambda t: (t[0][0], -t[0][3], t[0][5], t[0][1], t[0][2])):
        if kind not in kinds:
            continue
        oh, ow = ops.conv_out_hw(h, w, k, s)
        x = torch.randn(N, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, cout, oh, ow, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        wp, wb = ops.pack_weights(wt, k, s, 0, torch.bfloat16), ops.pack_weights(wt, k, s, 1, torch.bfloat16)
        acc = ops.bn_acc_new(cout, "cuda")
        y = torch.empty_like(dy)
        dw = torch.empty(cout, cin, k, k, device="cuda")
        if kind == "fwd":
            fn = lambda: ops.conv_fwd(x, wp, None, cout, k, s, acc, out=y)
        elif kind == "dgrad":
            fn = lambda: ops.conv_dgrad(dy, wb, cin, h, w, k, s)
        else:
            fn = lambda: ops.conv_wgrad(x, dy, k, s, torch.float32, out=dw)
        us = graph_time(fn)
        fl = 2.0 * N * oh * ow * cout * cin * k * k
        by = (x.numel() + dy.numel()) * 2
        ideal = max(fl / 2.5e15, by / 5e12) * 1e6
        t = tot.setdefault(kind, [0.0, 0.0, 0.0, 0])
        t[0] += us * cnt; t[1] += ideal * cnt; t[2] += fl * cnt; t[3] += cnt
        print(f"{kind:5s} x{cnt} {cin:4d}->{cout:4d} {h:3d}x{w:3d} k{k}s{s}  {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s {by / us / 1e3:6.0f} GB/s"
              f"  ideal {ideal:6.1f} us  x{us / ideal:5.2f}", flush=True)
        del x, dy, y
    for kind, (us, ideal, fl, n) in tot.items():
        print(f"TOTAL {kind}: {n} launches {us / 1e3:.3f} ms, ideal {ideal / 1e3:.3f} ms (x{us / ideal:.2f}), {fl / us / 1e6:.0f} TF/s = {fl / us / 1e6 / 2500:.3f} of MFMA peak")


if __name__ == "__main__":
    main()
