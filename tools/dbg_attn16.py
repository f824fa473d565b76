import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import blocks as ob
from oracle.params import det_fill_
from src.model.model_builder import Model
from src.hipops import ops
import emulated_ops as emu

dt = torch.float16
cfg = ob.PRESETS["n"]
img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(9)).cuda()
rec = {}
real_attn, real_dw = ops.attn_fwd, ops.dw_fwd
def attn(qkv, heads, dk, dh, scale):
    out = real_attn(qkv, heads, dk, dh, scale)
    rec.setdefault("attn", []).append((qkv.detach().clone(), heads, dk, dh, scale, [t.detach().clone() for t in out[:2]]))
    return out
def dw(x, w9, stats_acc=None):
    y = real_dw(x, w9, stats_acc)
    rec.setdefault("dw", []).append((x.detach().clone(), w9.detach().clone(), y.detach().clone()))
    return y
ops.attn_fwd, ops.dw_fwd = attn, dw
m = Model(**cfg, num_classes=80); det_fill_(m.state_dict(), 2); m = m.cuda().train()
with torch.autocast("cuda", dtype=dt):
    m(img)
qkv, heads, dk, dh, scale, (o, vp) = rec["attn"][0]
print("qkv absmax", float(qkv.float().abs().max()), "shape", tuple(qkv.shape), heads, dk, dh, scale)
o_ref, v_ref, _ = emu.attn_fwd(qkv.cpu(), heads, dk, dh, scale)
print("attn o   rel err", float((o.float().cpu() - o_ref.float()).abs().max() / o_ref.float().abs().max()), "finite", bool(torch.isfinite(o.float()).all()))
print("attn vp  equal", torch.equal(vp.cpu(), v_ref))
for i, (x, w9, y) in enumerate(rec["dw"]):
    y_ref = emu.dw_fwd(x.cpu(), w9.cpu())
    print(f"dw[{i}] {tuple(x.shape)} rel err", float((y.float().cpu() - y_ref.float()).abs().max() / y_ref.float().abs().max()), "x absmax", float(x.float().abs().max()))
