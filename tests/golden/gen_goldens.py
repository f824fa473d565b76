#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

Usage (from the repo root):  python tests/golden/gen_goldens.py [/root/reference]

The reference is imported read-only from its checkout; nothing of it is copied here: the .npz
files hold inputs and the reference's outputs only.  `torchvision` is absent from this image, so a
stub module satisfies the reference's `import torchvision` (src/utils/model_utils.py:4,
src/data/transforms.py:2); its single arithmetic use, torchvision.ops.nms (model_utils.py:264), is
bound to the oracle's greedy_nms -- which is why the NMS core is "parity unpinned" (oracle/__init__).
Parameters are overwritten with oracle.params.det_fill_ (a key-hashed deterministic stream) so that
fixtures need not carry weights.  torch version is recorded in each file.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True

from oracle.params import det_fill_            # noqa: E402
from oracle.postproc import greedy_nms         # noqa: E402

# ---- torchvision stub -----------------------------------------------------------------------
tv = types.ModuleType("torchvision")
tv.ops = types.ModuleType("torchvision.ops")
tv.ops.nms = lambda boxes, scores, thr: greedy_nms(boxes, scores, thr)
tv.transforms = types.ModuleType("torchvision.transforms")
tv.transforms.v2 = types.ModuleType("torchvision.transforms.v2")
for name, mod in (("torchvision", tv), ("torchvision.ops", tv.ops),
                  ("torchvision.transforms", tv.transforms), ("torchvision.transforms.v2", tv.transforms.v2)):
    sys.modules[name] = mod
sys.path.insert(0, REF)

from src.model import model_blocks as rb                     # noqa: E402
from src.model.losses import YoloDFLQFLoss                   # noqa: E402
from src.model.model_builder import Model                    # noqa: E402
from src.utils import model_utils as ru                      # noqa: E402
from src.training.train_model import decode_predictions      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
META = dict(torch_version=torch.__version__)
torch.set_num_threads(8)


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().float().numpy() if v.is_floating_point() else v.detach().numpy()
        flat[k] = np.asarray(v)
    flat["_torch_version"] = np.asarray(META["torch_version"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **flat)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in flat.values()) / 1024:.0f} KiB raw)")


def g(seed):
    return torch.Generator().manual_seed(seed)


# ---- 1. per-block forward/backward (train) and forward (eval) ----------------------------------
def block_case(tag, ctor, xshape, seed):
    m = torch.nn.ModuleDict({"m": ctor()})
    det_fill_(m.state_dict(), seed)
    x = torch.randn(*xshape, generator=g(100 + seed)).requires_grad_(True)
    m.train()
    y = m["m"](x)
    dy = torch.randn(y.shape, generator=g(200 + seed))
    y.backward(dy)
    arrs = {"x": x, "y_train": y, "dy": dy, "dx": x.grad}
    for k, p in m.named_parameters():
        if p.grad is not None:
            arrs["grad:" + k] = p.grad
    for k, b in m.named_buffers():
        arrs["buf:" + k] = b.clone()
    m.eval()
    with torch.no_grad():
        arrs["y_eval"] = m["m"](x.detach())
    save("block_" + tag, seed=seed, **arrs)


SiLU, Id = torch.nn.SiLU, torch.nn.Identity
block_case("conv3x3s2", lambda: rb.Conv(8, 16, SiLU(), k=3, s=2, p=1), (2, 8, 14, 14), 1)
block_case("conv1x1_id", lambda: rb.Conv(16, 24, Id()), (2, 16, 12, 12), 2)
block_case("convdw", lambda: rb.Conv(16, 16, SiLU(), k=3, p=1, g=16), (2, 16, 12, 12), 3)
block_case("residual", lambda: rb.Residual(16), (2, 16, 12, 12), 4)
block_case("c3k", lambda: rb.C3K(16, 16), (2, 16, 12, 12), 5)
block_case("c3k2_res", lambda: rb.C3K2(16, 32, 1, False, 4), (2, 16, 12, 12), 6)
block_case("c3k2_csp", lambda: rb.C3K2(32, 32, 2, True, 2), (2, 32, 12, 12), 7)
block_case("sppf", lambda: rb.SPPF(16, 16), (2, 16, 12, 12), 8)
block_case("attention", lambda: rb.Attention(128, 2), (2, 128, 6, 6), 9)
block_case("psablock", lambda: rb.PSABlock(128, 2), (2, 128, 6, 6), 10)
block_case("psa", lambda: rb.PSA(256, 1), (2, 256, 6, 6), 11)

x = torch.randn(3, 64, 50, generator=g(12))
save("block_dfl", x=x, y=rb.DFL(16)(x))

# ---- 2. make_anchors (fp32 and bf16), dist2bbox, xywh2xyxy, box_iou -----------------------------
feat = [torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 2, 2)]
a32, s32 = ru.make_anchors(feat, [8.0, 16.0, 32.0], 0.5)
feat16 = [torch.zeros(1, 1, 160, 160, dtype=torch.bfloat16), torch.zeros(1, 1, 80, 80, dtype=torch.bfloat16)]
a16, s16 = ru.make_anchors(feat16, [8.0, 16.0], 0.5)
d = torch.rand(2, 4, 84, generator=g(13)) * 6
bx = torch.rand(40, 4, generator=g(14)) * 100
b1 = ru.xywh2xyxy(bx[:20])
b2 = ru.xywh2xyxy(bx[20:])
save("utils", anchors32=a32, strides32=s32, anchors_bf16=a16.float(), strides_bf16=s16.float(),
     dist=d, d2b_xywh=ru.dist2bbox(d, a32.t().unsqueeze(0), xywh=True, dim=1),
     d2b_xyxy=ru.dist2bbox(d, a32.t().unsqueeze(0), xywh=False, dim=1),
     xywh=bx, xyxy=ru.xywh2xyxy(bx), box_iou=ru.box_iou(b1, b2))

# ---- 3. loss: tiny hand-built case (empty image, duplicate assignment, quirk IoU > 0) ----------
A, NC = 84, 8
anc, st = a32.t().contiguous(), s32.t().contiguous()          # (2,84), (1,84) as Head returns them
preds = torch.randn(3, 64 + NC, A, generator=g(20))
preds[:, 64:] -= 2.0
gts = [
    torch.tensor([[20., 20., 16., 12., 3.], [21., 20.5, 10., 30., 5.],      # two GTs -> same anchor
                  [50., 9., 8., 6., 1.], [40., 44., 30., 20., 7.]]),
    torch.zeros(0, 5),                                                      # image without GT
    torch.tensor([[12., 30., 9., 70., 2.], [55., 60., 20., 18., 0.],        # tall box: quirk IoU>0
                  [33., 10., 64., 8., 6.], [5., 5., 4., 4., 4.], [60., 30., 40., 100., 3.]]),
]
crit = YoloDFLQFLoss(num_classes=NC, lambda_box=1.5, lambda_cls=1.0)
p = preds.clone().requires_grad_(True)
loss, ld = crit(p, gts, anc, st)
loss.backward()
save("loss_small", preds=preds, anchors=anc, strides=st, gt0=gts[0], gt1=gts[1], gt2=gts[2],
     total=ld["total_loss"], box=ld["box_loss"], cls=ld["cls_loss"], dpreds=p.grad)

# lambda variants (lambda_box must have no effect; lambda_cls/lambda_dfl scale)
crit2 = YoloDFLQFLoss(num_classes=NC, lambda_box=7.0, lambda_cls=0.5, lambda_dfl=2.0)
p = preds.clone().requires_grad_(True)
loss2, ld2 = crit2(p, gts, anc, st)
loss2.backward()
save("loss_small_lambdas", total=ld2["total_loss"], box=ld2["box_loss"], cls=ld2["cls_loss"], dpreds=p.grad)

# bf16 preds + bf16 anchors (what DDP-autocast hands the loss)
pb = preds.to(torch.bfloat16)
p = pb.clone().requires_grad_(True)
loss3, ld3 = crit(p, gts, anc.to(torch.bfloat16), st.to(torch.bfloat16))
loss3.backward()
save("loss_small_bf16", total=ld3["total_loss"], box=ld3["box_loss"], cls=ld3["cls_loss"],
     dpreds=p.grad.float())

# ---- 4. loss at nano@320 size: random preds, random COCO-shaped targets ------------------------
fe = [torch.zeros(1, 1, 40, 40), torch.zeros(1, 1, 20, 20), torch.zeros(1, 1, 10, 10)]
aL, sL = ru.make_anchors(fe, [8.0, 16.0, 32.0], 0.5)
aL, sL = aL.t().contiguous(), sL.t().contiguous()
gg = g(30)
predsL = torch.randn(2, 144, 2100, generator=gg)
predsL[:, 64:] = predsL[:, 64:] * 0.5 - 4.0
gtL = []
for i in range(2):
    m = 7 + 6 * i
    c = torch.rand(m, 2, generator=gg) * 320
    wh = torch.rand(m, 2, generator=gg) * (0.4 * 320) + 8
    cl = torch.randint(0, 80, (m, 1), generator=gg).float()
    gtL.append(torch.cat([c, wh, cl], 1))
critL = YoloDFLQFLoss(num_classes=80)
p = predsL.clone().requires_grad_(True)
lossL, ldL = critL(p, gtL, aL, sL)
lossL.backward()
gsel = p.grad[:, :, ::25].contiguous()
save("loss_n320", seed=30, gt0=gtL[0], gt1=gtL[1], total=ldL["total_loss"], box=ldL["box_loss"],
     cls=ldL["cls_loss"], dpreds_sum=p.grad.double().sum(), dpreds_abs=p.grad.double().abs().sum(),
     dpreds_stride25=gsel)

# ---- 5. decode (validation) and inference decode + NMS -----------------------------------------
gg = g(40)
pd = torch.randn(2, 64 + NC, A, generator=gg)
pd[:, 64:] += 0.5
dec = decode_predictions(pd, anc, st, conf_threshold=0.6, top_k=10, num_classes=NC)
save("decode_val", preds=pd, anchors=anc, strides=st, out0=dec[0], out1=dec[1])

gg = g(41)
M = 512
pn = torch.empty(2, 4 + NC, M)
pn[:, 0:2] = torch.rand(2, 2, M, generator=gg) * 300
pn[:, 2:4] = torch.rand(2, 2, M, generator=gg) * 80 + 10
pn[:, 4:] = torch.rand(2, NC, M, generator=gg)            # distinct scores w.p. 1
variants = {
    "default": dict(conf_thres=0.25, iou_thres=0.45),
    "agnostic": dict(conf_thres=0.25, iou_thres=0.45, agnostic=True),
    "multi": dict(conf_thres=0.6, iou_thres=0.5, multi_label=True),
    "classes": dict(conf_thres=0.25, iou_thres=0.45, classes=[1, 3, 6]),
    "maxdet": dict(conf_thres=0.25, iou_thres=0.9, max_det=17),
    "highconf": dict(conf_thres=0.999, iou_thres=0.45),
}
arrs = {"prediction": pn}
for tag, kw in variants.items():
    res = ru.non_max_suppression(pn.clone(), nc=NC, **kw)
    for i, r in enumerate(res):
        arrs[f"{tag}:{i}"] = r
save("nms", **arrs)

# ---- 6. full model, nano@320 N=2: forward, loss, a few gradients, fuse(), inference ------------
cfg = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
model = Model(**cfg, num_classes=80)
det_fill_(model.state_dict(), 0)
gg = g(50)
img = torch.randn(2, 3, 320, 320, generator=gg)
model.train()
preds, an, stt = model(img)
crit80 = YoloDFLQFLoss(num_classes=80)
loss, ld = crit80(preds, gtL, an, stt)
loss.backward()
grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
keys = ["net.p1.0.conv.weight", "net.p1.0.norm.weight", "net.p2.1.conv2.conv.weight",
        "net.p5.3.res_m.0.conv1.qkv.conv.weight", "net.p5.3.res_m.0.conv1.conv1.conv.weight",
        "fpn.h2.conv1.conv.weight", "head.box.0.2.weight", "head.box.0.2.bias",
        "head.cls.2.4.weight", "head.cls.1.0.conv.weight", "head.cls.0.3.norm.bias"]
arrs = {"grad:" + k: grads[k] for k in keys}
arrs["gradnorms_keys"] = np.asarray(sorted(grads))
arrs["gradnorms"] = np.asarray([float(grads[k].double().norm()) for k in sorted(grads)])
sd = model.state_dict()
arrs["rm:net.p1.0"] = sd["net.p1.0.norm.running_mean"]
arrs["rv:net.p1.0"] = sd["net.p1.0.norm.running_var"]
arrs["rv:head.cls.2.3"] = sd["head.cls.2.3.norm.running_var"]
save("model_n320_train", seed=0, preds_stride7=preds[:, :, ::7].contiguous(), preds_sum=preds.double().sum(),
     preds_abs=preds.double().abs().sum(), anchors=an, strides=stt, total=ld["total_loss"],
     box=ld["box_loss"], cls=ld["cls_loss"], **arrs)

det_fill_(model.state_dict(), 0)     # undo the running-stat update of the train pass
model.eval()
with torch.no_grad():
    pe, _, _ = model(img)
    dets = model.inference(img, conf_thres=0.0, iou_thres=0.45)
    model.fuse()
    pf, _, _ = model(img)
arrs = {f"det:{i}": d_ for i, d_ in enumerate(dets)}
save("model_n320_eval", seed=0, preds_stride7=pe[:, :, ::7].contiguous(), preds_abs=pe.double().abs().sum(),
     fused_stride7=pf[:, :, ::7].contiguous(), fused_key_sample=sd["net.p1.0.norm.weight"], **arrs)
print("done")
