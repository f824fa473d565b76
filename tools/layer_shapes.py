"""Distinct dense-conv calls (forward / data gradient / weight gradient) of one training step of a preset, enumerated on the
CPU by running the model once at 64x64 with the torch stand-ins of tests/emulated_ops.py and scaling the maps.
Used by tools/conv_layers.py (per-layer timing against the layer's own roofline)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "custom-yolo-implmentation_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def conv_calls(preset="s", res=640):
    """-> Counter{(kind, cin, cout, h, w, k, stride): count}; h, w = INPUT map of the forward conv."""
    import torch
    import emulated_ops as emu
    from oracle import blocks as ob
    from src.hipops import ops as o
    saved = {name: getattr(o, name) for name in emu.LEAVES}
    emu.install_plain()
    calls = []
    rf, rd, rw = o.conv_fwd, o.conv_dgrad, o.conv_wgrad

    def fwd(x, wp, bias, cout, k, stride, stats_acc=None, out=None):
        calls.append(("fwd", x.shape[1], cout, x.shape[2], x.shape[3], k, stride))
        return rf(x, wp, bias, cout, k, stride, stats_acc, out)

    def dgrad(dy, wb, cin, h, w, k, stride, acc_into=None, acc2=None):
        calls.append(("dgrad", cin, dy.shape[1], h, w, k, stride))
        return rd(dy, wb, cin, h, w, k, stride, acc_into, acc2)

    def wgrad(x, dy, k, stride, w_dtype, out=None):
        calls.append(("wgrad", x.shape[1], dy.shape[1], x.shape[2], x.shape[3], k, stride))
        r = rw(x, dy, k, stride, w_dtype)
        return r if out is None else out.copy_(r)

    o.conv_fwd, o.conv_dgrad, o.conv_wgrad = fwd, dgrad, wgrad
    try:
        from src.model.model_builder import Model
        m = Model(**ob.PRESETS[preset], num_classes=80).train()
        preds, _, _ = m(torch.randn(1, 3, 64, 64))
        preds.sum().backward()
    finally:
        for name, fn in saved.items():
            setattr(o, name, fn)
        o.conv_fwd, o.conv_dgrad, o.conv_wgrad = saved["conv_fwd"], saved["conv_dgrad"], saved["conv_wgrad"]
    sc = res // 64
    out = collections.Counter()
    for kind, cin, cout, h, w, k, s in calls:
        if cin >= 8:
            out[(kind, cin, cout, h * sc, w * sc, k, s)] += 1
    return out


if __name__ == "__main__":
    for key, cnt in sorted(conv_calls(*sys.argv[1:2]).items()):
        print(cnt, key)
