"""Sweep of the ring conv kernel's (K-step, pixel tile, channel tile, depth) per layer shape of a preset against the
gather / halo kernels' own choice; graph-replayed launches, warm clocks.  Usage: ring_tune.py [preset] [fwd|dgrad ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from conv_layers import graph_time
from layer_shapes import conv_calls
from src.hipops import lib, ops

N = 32


def tune(bn=0, tap_inner=-1, halo=-1, dma=-1, ring=-1, bm=0, nst=0, bk=0):
    lib.call("yolo_conv_tune_set", bn, tap_inner, halo, dma, ring, bm, nst, bk)


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "s"
    kinds = set(sys.argv[2:]) or {"fwd", "dgrad"}
    tot_old = tot_best = 0.0
    for (kind, cin, cout, h, w, k, s), cnt in sorted(conv_calls(preset).items(), key=lambda t: (t[0][0], -t[0][3], t[0][5], t[0][1], t[0][2])):
        if kind not in kinds:
            continue
        oh, ow = ops.conv_out_hw(h, w, k, s)
        x = torch.randn(N, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, cout, oh, ow, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        wp, wb = ops.pack_weights(wt, k, s, 0, torch.bfloat16), ops.pack_weights(wt, k, s, 1, torch.bfloat16)
        acc = ops.bn_acc_new(cout, "cuda")
        y = torch.empty_like(dy)
        fn = (lambda: ops.conv_fwd(x, wp, None, cout, k, s, acc, out=y)) if kind == "fwd" else (lambda: ops.conv_dgrad(dy, wb, cin, h, w, k, s))
        tune(ring=0)
        old = graph_time(fn)
        cs, cd = (cin, cout) if kind == "fwd" else (cout, cin)
        res = []
        if cs % 32 == 0 and cs >= 64:
            for bk in (32, 64):
                for bm in (128, 64):
                    for bn in (128, 64, 32):
                        if bn > max(64, cd) or (bn == 32 and (bk == 32 or cd > 32)):
                            continue
                        for nst in (2, 3, 4):
                            if nst * (bm + bn) * bk * 2 > 160 * 1024:
                                continue
                            tune(bn=bn, halo=0, ring=1, bm=bm, nst=nst, bk=bk)
                            try:
                                res.append((graph_time(fn), f"k{bk}/{bm}x{bn}/{nst}"))
                            except Exception as e:
                                res.append((1e9, f"k{bk}/{bm}x{bn}/{nst}:ERR"))
        res.sort()
        best = min(old, res[0][0]) if res else old
        tot_old += old * cnt
        tot_best += best * cnt
        fl = 2.0 * N * oh * ow * cout * cin * k * k
        by = (x.numel() + dy.numel()) * 2
        ideal = max(fl / 2.5e15, by / 5e12) * 1e6
        print(f"{kind:5s} x{cnt} {cin:4d}->{cout:4d} {h:3d}x{w:3d} k{k}s{s} ideal {ideal:6.1f} | old {old:7.1f} | " +
              "  ".join(f"{u:6.1f}@{c}" for u, c in res[:5]), flush=True)
        del x, dy, y
    print(f"TOTAL old {tot_old / 1e3:.3f} ms, best-of {tot_best / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
