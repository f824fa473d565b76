"""Pin the CPU oracle against vectors produced by the reference itself (tests/golden/gen_goldens.py).

Tolerance: fp32 restatement vs reference <= 1e-6 abs / 1e-5 rel on outputs, loss and
gradients; NMS rows exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import blocks as ob
from oracle import loss as ol
from oracle import postproc as op
from oracle.params import ParamStore

RTOL, ATOL = 1e-5, 2e-6


def close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    assert bool((err <= lim).all()), f"{what}: max err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"


BLOCKS = {
    "conv3x3s2": lambda ps, x, c: ob.conv_block(ps, "m", x, 8, 16, True, c, k=3, s=2, p=1),
    "conv1x1_id": lambda ps, x, c: ob.conv_block(ps, "m", x, 16, 24, False, c),
    "convdw": lambda ps, x, c: ob.conv_block(ps, "m", x, 16, 16, True, c, k=3, p=1, g=16),
    "residual": lambda ps, x, c: ob.residual(ps, "m", x, 16, c),
    "c3k": lambda ps, x, c: ob.c3k(ps, "m", x, 16, 16, c),
    "c3k2_res": lambda ps, x, c: ob.c3k2(ps, "m", x, 16, 32, 1, False, 4, c),
    "c3k2_csp": lambda ps, x, c: ob.c3k2(ps, "m", x, 32, 32, 2, True, 2, c),
    "sppf": lambda ps, x, c: ob.sppf(ps, "m", x, 16, 16, c),
    "attention": lambda ps, x, c: ob.attention(ps, "m", x, 128, 2, c),
    "psablock": lambda ps, x, c: ob.psablock(ps, "m", x, 128, 2, c),
    "psa": lambda ps, x, c: ob.psa(ps, "m", x, 256, 1, c),
}


@pytest.mark.parametrize("tag", sorted(BLOCKS))
def test_block_matches_reference(tag):
    gd = load_golden("block_" + tag)
    ps = ParamStore(int(gd["seed"]), requires_grad=True)
    x = gd["x"].clone().requires_grad_(True)
    y = BLOCKS[tag](ps, x, ob.Ctx(training=True))
    close(y, gd["y_train"], what="y_train")
    y.backward(gd["dy"])
    close(x.grad, gd["dx"], rtol=1e-4, atol=1e-5, what="dx")
    for k, v in gd.items():
        if k.startswith("grad:"):
            close(ps[k[5:]].grad, v, rtol=1e-4, atol=2e-5, what=k)
        if k.startswith("buf:"):
            close(ps[k[4:]], v, what=k)
    with torch.no_grad():   # eval after the train pass: uses the running stats that pass updated
        close(BLOCKS[tag](ps, gd["x"], ob.Ctx(training=False)), gd["y_eval"], what="y_eval")


def test_dfl_block():
    gd = load_golden("block_dfl")
    close(ob.dfl(ParamStore(), gd["x"]), gd["y"])


def test_anchors_and_box_utils():
    gd = load_golden("utils")
    a, s = ob.make_anchors([(8, 8), (4, 4), (2, 2)], [8.0, 16.0, 32.0])
    assert torch.equal(a, gd["anchors32"]) and torch.equal(s, gd["strides32"])
    a, s = ob.make_anchors([(160, 160), (80, 80)], [8.0, 16.0], dtype=torch.bfloat16)
    assert torch.equal(a.float(), gd["anchors_bf16"]) and torch.equal(s.float(), gd["strides_bf16"])
    assert torch.equal(op.xywh2xyxy(gd["xywh"]), gd["xyxy"])


def _loss_inputs():
    gd = load_golden("loss_small")
    return gd, [gd["gt0"], gd["gt1"].reshape(0, 5), gd["gt2"]]


def test_loss_small_value_and_grad():
    gd, gts = _loss_inputs()
    p = gd["preds"].clone().requires_grad_(True)
    tot, box, cls = ol.dfl_qfl_loss(p, gts, gd["anchors"], gd["strides"], 8)
    close(tot, gd["total"]), close(box, gd["box"]), close(cls, gd["cls"])
    tot.backward()
    close(p.grad, gd["dpreds"], rtol=1e-4, atol=1e-7, what="dpreds")


def test_loss_lambdas():
    gd, gts = _loss_inputs()
    g2 = load_golden("loss_small_lambdas")
    p = gd["preds"].clone().requires_grad_(True)
    tot, box, cls = ol.dfl_qfl_loss(p, gts, gd["anchors"], gd["strides"], 8, lambda_cls=0.5, lambda_dfl=2.0)
    close(tot, g2["total"]), close(box, g2["box"]), close(cls, g2["cls"])
    tot.backward()
    close(p.grad, g2["dpreds"], rtol=1e-4, atol=1e-7)


def test_loss_bf16_inputs():
    gd, gts = _loss_inputs()
    g3 = load_golden("loss_small_bf16")
    p = gd["preds"].to(torch.bfloat16).requires_grad_(True)
    tot, box, cls = ol.dfl_qfl_loss(p, gts, gd["anchors"].bfloat16(), gd["strides"].bfloat16(), 8)
    close(tot, g3["total"]), close(box, g3["box"]), close(cls, g3["cls"])
    tot.backward()
    close(p.grad.float(), g3["dpreds"], rtol=1e-2, atol=1e-6)   # grad itself is rounded to bf16


def n320_inputs():
    gd = load_golden("loss_n320")
    gg = torch.Generator().manual_seed(int(gd["seed"]))
    preds = torch.randn(2, 144, 2100, generator=gg)
    preds[:, 64:] = preds[:, 64:] * 0.5 - 4.0
    a, s = ob.make_anchors([(40, 40), (20, 20), (10, 10)], [8.0, 16.0, 32.0])
    return gd, preds, [gd["gt0"], gd["gt1"]], a.t().contiguous(), s.t().contiguous()


def test_loss_n320():
    gd, preds, gts, a, s = n320_inputs()
    p = preds.clone().requires_grad_(True)
    tot, box, cls = ol.dfl_qfl_loss(p, gts, a, s, 80)
    close(tot, gd["total"]), close(box, gd["box"]), close(cls, gd["cls"])
    tot.backward()
    close(p.grad[:, :, ::25], gd["dpreds_stride25"], rtol=1e-4, atol=1e-9)
    close(p.grad.double().abs().sum(), gd["dpreds_abs"], rtol=1e-5)


def test_decode_val():
    gd = load_golden("decode_val")
    out = op.decode_predictions(gd["preds"], gd["anchors"], gd["strides"], conf_threshold=0.6, top_k=10)
    for i in range(2):
        close(out[i], gd[f"out{i}"].reshape(-1, 5), what=f"decode{i}")


NMS_VARIANTS = {
    "default": dict(conf_thres=0.25, iou_thres=0.45),
    "agnostic": dict(conf_thres=0.25, iou_thres=0.45, agnostic=True),
    "multi": dict(conf_thres=0.6, iou_thres=0.5, multi_label=True),
    "classes": dict(conf_thres=0.25, iou_thres=0.45, classes=[1, 3, 6]),
    "maxdet": dict(conf_thres=0.25, iou_thres=0.9, max_det=17),
    "highconf": dict(conf_thres=0.999, iou_thres=0.45),
}


@pytest.mark.parametrize("tag", sorted(NMS_VARIANTS))
def test_nms_pipeline(tag):
    gd = load_golden("nms")
    out = op.non_max_suppression(gd["prediction"].clone(), nc=8, **NMS_VARIANTS[tag])
    for i in range(2):
        want = gd[f"{tag}:{i}"]
        want = want.reshape(-1, 6) if isinstance(want, torch.Tensor) else torch.zeros(0, 6)
        assert out[i].shape == want.shape, (tag, i, out[i].shape, want.shape)
        assert torch.equal(out[i], want), (tag, i)


def _nano():
    p = ob.PRESETS["n"]
    return p["width"], p["depth"], p["csp"]


def test_model_n320_train_forward_loss_grads():
    gd = load_golden("model_n320_train")
    l3 = load_golden("loss_n320")
    w, d, c = _nano()
    ps = ParamStore(int(gd["seed"]), requires_grad=True)
    img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(50))
    preds, a, s = ob.model_forward(ps, img, w, d, c, 80, training=True)
    close(preds[:, :, ::7], gd["preds_stride7"], rtol=1e-4, atol=1e-5, what="preds")
    assert torch.equal(a, gd["anchors"]) and torch.equal(s, gd["strides"])
    tot, box, cls = ol.dfl_qfl_loss(preds, [l3["gt0"], l3["gt1"]], a, s, 80)
    close(tot, gd["total"], rtol=1e-5), close(box, gd["box"], rtol=1e-5), close(cls, gd["cls"], rtol=1e-5)
    tot.backward()
    for k, v in gd.items():
        if k.startswith("grad:"):
            g = ps[k[5:]].grad
            scale = float(v.abs().max())
            close(g, v, rtol=1e-3, atol=1e-4 * scale + 1e-9, what=k)
    keys = [str(k) for k in gd["gradnorms_keys"]]
    norms = torch.as_tensor(gd["gradnorms"])
    mine = torch.tensor([float(ps[k].grad.double().norm()) for k in keys])
    close(mine, norms, rtol=2e-3, atol=1e-9, what="grad norms of every parameter")
    close(ps["net.p1.0.norm.running_mean"], gd["rm:net.p1.0"])
    close(ps["net.p1.0.norm.running_var"], gd["rv:net.p1.0"])
    close(ps["head.cls.2.3.norm.running_var"], gd["rv:head.cls.2.3"])


def test_model_n320_eval_fuse_inference():
    gd = load_golden("model_n320_eval")
    w, d, c = _nano()
    ps = ParamStore(int(gd["seed"]))
    img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(50))
    with torch.no_grad():
        preds, a, s = ob.model_forward(ps, img, w, d, c, 80, training=False)
        close(preds[:, :, ::7], gd["preds_stride7"], rtol=1e-4, atol=1e-5)
        fused = ob.fuse_state(ps)
        pf, _, _ = ob.model_forward(fused, img, w, d, c, 80, training=False, fused=True)
        close(pf[:, :, ::7], gd["fused_stride7"], rtol=1e-4, atol=2e-5)
        y = op.inference_decode(ps, preds, a, s, 80)
        dets = op.non_max_suppression(y, conf_thres=0.0, iou_thres=0.45, nc=80)
    for i, dt in enumerate(dets):
        want = gd[f"det:{i}"]
        want = want.reshape(-1, 6) if isinstance(want, torch.Tensor) else torch.zeros(0, 6)
        assert dt.shape == want.shape, (i, dt.shape, want.shape)
        close(dt, want, rtol=1e-4, atol=1e-3, what=f"det{i}")
        assert torch.equal(dt[:, 5], want[:, 5])


def _metric_cases(gd):
    po, go = gd["pred_off"], gd["gt_off"]
    return [(gd["pred"][po[i]:po[i + 1]], gd["gt"][go[i]:go[i + 1]]) for i in range(len(po) - 1)]


@pytest.mark.parametrize("thr", [0.5, 0.45])
def test_detection_metrics_counters(thr):
    """oracle MetricCounters vs the reference's DetectionMetrics (metrics.py:68-191): exact counters"""
    gd = load_golden("metrics")
    cases = _metric_cases(gd)
    close(op.box_iou_batch(cases[0][0][:, :4], cases[0][1][:, :4]), gd["iou_first"], rtol=0, atol=0, what="iou")
    m = op.MetricCounters(int(gd["num_classes"]), thr)
    tag = f"thr{thr}"
    for i, (p, t) in enumerate(cases):
        m.update(p, t)
        assert m.scalars() == gd[tag + ":scalars_after_each"][i].tolist(), f"image {i}"
    for name, arr in (("class_tp", m.class_tp), ("class_fp", m.class_fp), ("class_fn", m.class_fn), ("class_gt", m.class_gt)):
        assert arr.tolist() == gd[f"{tag}:{name}"].long().tolist(), name
    res = m.compute()
    for k, v in zip(gd[tag + ":compute_keys"], gd[tag + ":compute_vals"].tolist()):
        assert abs(res[str(k)] - v) <= 1e-12 + 1e-7 * abs(v), (k, res[str(k)], v)
