"""-m gpu: whole-path parity.  The product model + loss on the MI355X against (a) vectors the reference
produced (tests/golden) and (b) the CPU oracle on the same seeded inputs.

fp32 path: 1e-3 relative (north_star's bound) on outputs / loss / gradients -- measured errors are ~1e-5.
bf16 autocast path (what the bench runs): activations are rounded to bf16 after every layer exactly as
the reference's autocast does; the loss scalars must stay within 2e-2 of the fp32 oracle and the test
prints the measured error so DESIGN.md can quote it."""
import pytest
import torch

from conftest import load_golden
from oracle import blocks as ob
from oracle import loss as ol
from oracle.params import ParamStore, det_fill_

pytestmark = pytest.mark.gpu
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _model(seed=0, cfg=NANO, nc=80):
    from src.model.model_builder import Model
    m = Model(**cfg, num_classes=nc)
    det_fill_(m.state_dict(), seed)
    return m.cuda()


def test_config1_nano320_fp32_matches_reference_goldens():
    from src.model.losses import YoloDFLQFLoss
    gd, l3 = load_golden("model_n320_train"), load_golden("loss_n320")
    model = _model().train()
    img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(50))
    preds, a, s = model(img.cuda())
    assert _rel(preds[:, :, ::7], gd["preds_stride7"]) < 1e-3
    loss, ld = YoloDFLQFLoss(num_classes=80)(preds, [l3["gt0"].cuda(), l3["gt1"].cuda()], a, s)
    for k, g in (("total_loss", "total"), ("box_loss", "box"), ("cls_loss", "cls")):
        assert abs(ld[k] - float(gd[g])) <= 1e-3 * abs(float(gd[g])), (k, ld[k], float(gd[g]))
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    worst = 0.0
    for k, v in gd.items():
        if k.startswith("grad:"):
            worst = max(worst, _rel(grads[k[5:]], v))
    keys = [str(k) for k in gd["gradnorms_keys"]]
    mine = torch.tensor([float(grads[k].double().norm()) for k in keys])
    ref = torch.as_tensor(gd["gradnorms"])
    big = ref > 1e-4 * ref.max()
    nerr = float(((mine - ref).abs() / ref)[big].max())
    print(f"\n[parity fp32 nano@320] preds {_rel(preds[:, :, ::7], gd['preds_stride7']):.2e} "
          f"loss {abs(ld['total_loss'] - float(gd['total'])) / float(gd['total']):.2e} grad(max of 11 tensors) {worst:.2e} "
          f"grad-norms(all {len(keys)}) {nerr:.2e}")
    assert worst < 5e-3 and nerr < 5e-3
    sd = model.state_dict()
    assert _rel(sd["net.p1.0.norm.running_var"], gd["rv:net.p1.0"]) < 1e-4


def test_eval_fuse_inference_matches_reference_goldens():
    gd = load_golden("model_n320_eval")
    model = _model().eval()
    img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(50)).cuda()
    with torch.no_grad():
        preds, _, _ = model(img)
        assert _rel(preds[:, :, ::7], gd["preds_stride7"]) < 1e-3
        dets = model.inference(img, conf_thres=0.0, iou_thres=0.45)
        model.fuse()
        pf, _, _ = model(img)
        assert _rel(pf[:, :, ::7], gd["fused_stride7"]) < 1e-3
    for i, dt in enumerate(dets):
        want = gd[f"det:{i}"]
        want = want.reshape(-1, 6) if isinstance(want, torch.Tensor) else torch.zeros(0, 6)
        assert dt.shape == want.shape
        if want.numel():
            assert torch.equal(dt[:, 5].cpu(), want[:, 5]) and _rel(dt[:, :5], want[:, :5]) < 1e-3


@pytest.mark.parametrize("res,n", [(160, 2), (320, 2)])
def test_bf16_autocast_step_vs_fp32_oracle(res, n):
    from src.model.losses import YoloDFLQFLoss
    model = _model(seed=3).train()
    g = torch.Generator().manual_seed(7)
    img = torch.randn(n, 3, res, res, generator=g)
    gts = [torch.cat([torch.rand(3 + i, 2, generator=g) * res, torch.rand(3 + i, 2, generator=g) * res * 0.4 + 8,
                      torch.randint(0, 80, (3 + i, 1), generator=g).float()], 1) for i in range(n)]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        preds, a, s = model(img.cuda())
        loss, ld = YoloDFLQFLoss(num_classes=80)(preds, [t.cuda() for t in gts], a, s)
    assert preds.dtype == torch.bfloat16
    loss.backward()
    ps = ParamStore(3, requires_grad=True)
    p_ref, a_ref, s_ref = ob.model_forward(ps, img, NANO["width"], NANO["depth"], NANO["csp"], 80, training=True)
    tot, dfl, cls = ol.dfl_qfl_loss(p_ref, gts, a_ref, s_ref, 80)
    tot.backward()
    # the same restatement under CPU bf16 autocast = what the reference's own bf16 path computes
    ps16 = ParamStore(3)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        p16, a16, s16 = ob.model_forward(ps16, img, NANO["width"], NANO["depth"], NANO["csp"], 80, training=True)
        t16, d16, c16 = ol.dfl_qfl_loss(p16, gts, a16, s16, 80)
    e_tot = abs(ld["total_loss"] - float(tot)) / float(tot)
    e_cls = abs(ld["cls_loss"] - float(cls)) / float(cls)
    e_dfl = abs(ld["box_loss"] - float(dfl)) / float(dfl)
    r_tot, r_cls, r_dfl = abs(float(t16 - tot)) / float(tot), abs(float(c16 - cls)) / float(cls), abs(float(d16 - dfl)) / float(dfl)
    gk = "head.cls.0.4.weight"
    e_g = _rel(dict(model.named_parameters())[gk].grad, ps[gk].grad)
    print(f"\n[parity bf16 nano@{res}] HIP-bf16 vs fp32 oracle: total {e_tot:.2e} dfl {e_dfl:.2e} cls {e_cls:.2e} "
          f"preds {_rel(preds, p_ref):.2e} grad {e_g:.2e} | CPU-bf16-autocast vs fp32 oracle: total {r_tot:.2e} "
          f"dfl {r_dfl:.2e} cls {r_cls:.2e} preds {_rel(p16, p_ref):.2e}")
    # bf16 rounding noise is inherent to the precision (the reference's own CPU bf16 path moves preds by
    # ~1e-1 max-rel here).  The loss is a chaotic function of it at this size -- one GT re-assigned to a
    # neighbouring anchor moves a mean over ~7 positives by >10 % -- and the batch statistics are summed with float
    # atomics, so the HIP value moves a little from run to run: (1) the predictions must stay in the band of the
    # CPU bf16 path, (2) the loss kernels must agree tightly with the oracle's loss evaluated ON THE SAME
    # predictions, (3) the cross-precision loss only gets a sanity band.
    assert _rel(preds, p_ref) < 2 * _rel(p16, p_ref) + 1e-2
    with torch.no_grad():
        t_same, d_same, c_same = ol.dfl_qfl_loss(preds.detach().float().cpu(), gts, a.float().cpu(), s.float().cpu(), 80)
    assert abs(ld["total_loss"] - float(t_same)) / float(t_same) < 5e-3, (ld["total_loss"], float(t_same))
    assert abs(ld["cls_loss"] - float(c_same)) / float(c_same) < 5e-3 and abs(ld["box_loss"] - float(d_same)) / float(d_same) < 5e-3
    for e, r in ((e_tot, r_tot), (e_dfl, r_dfl), (e_cls, r_cls)):
        assert e < max(4 * r, 0.2), (e, r)


def test_batched_weight_packing_equals_per_layer_packing():
    """Model.prepack (one launch for all conv weights) must reproduce the per-layer packed matrices bit for bit,
    and a parameter update must invalidate the served copies."""
    from src.hipops import ops
    model = _model(seed=5).train()
    model.prepack = True
    img = torch.randn(2, 3, 96, 96, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(img)
    plan = ops.ACTIVE_PACK_PLAN
    assert plan is not None and plan.njobs > 100
    n = 0
    for (ptr, mode), (view, w, ver) in plan.entries.items():
        o, i, k = w.shape[0], w.shape[1], w.shape[2]
        s = 2 if any(m.conv.weight is w and m._s == 2 for m in model.modules() if hasattr(m, "_s")) else 1
        ops.ACTIVE_PACK_PLAN = None
        ref = ops.pack_weights(w, k, s, mode, torch.bfloat16)
        ops.ACTIVE_PACK_PLAN = plan
        assert torch.equal(view, ref), (tuple(w.shape), mode)
        assert ops.pack_weights(w, k, s, mode, torch.bfloat16).data_ptr() == view.data_ptr()
        n += 1
    w0 = model.net.p2[0].conv.weight
    with torch.no_grad():
        w0.add_(1.0)                                   # version bump -> the plan must refuse to serve it
    assert plan.lookup(w0, 0, torch.bfloat16) is None
    ops.ACTIVE_PACK_PLAN = None
    assert n > 100


def test_small_preset_640_bf16_step_runs_and_is_finite():
    """BASELINE config 2's model at a reduced batch: shapes, finiteness, every parameter gets a gradient."""
    from src.model.losses import YoloDFLQFLoss
    model = _model(cfg=ob.PRESETS["s"]).train()
    g = torch.Generator().manual_seed(11)
    img = torch.randn(2, 3, 640, 640, generator=g).cuda()
    gts = [torch.tensor([[320., 300., 100., 80., 4.], [100., 500., 60., 90., 33.]]).cuda(), torch.zeros(0, 5).cuda()]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        preds, a, s = model(img)
        loss, ld = YoloDFLQFLoss(num_classes=80)(preds, gts, a, s)
    loss.backward()
    assert preds.shape == (2, 144, 8400) and torch.isfinite(preds.float()).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for k, p in model.named_parameters()
               if k != "head.dfl.conv.weight")
    assert ld["total_loss"] > 0


def _grad_errors(grads, ps):
    """{key: (cosine, relative L2)} of `grads[key]` against the fp32 oracle's ps[key].grad (near-zero gradients skipped)."""
    gmax = max(float(v.grad.norm()) for k, v in ps.items() if getattr(v, "grad", None) is not None)
    out = {}
    for k, v in ps.items():
        if getattr(v, "grad", None) is None or float(v.grad.norm()) < 1e-6 * gmax:
            continue
        g, r = grads[k].detach().double().cpu().flatten(), v.grad.double().flatten()
        assert torch.isfinite(g).all(), k
        out[k] = (float(torch.dot(g, r) / (g.norm() * r.norm()).clamp_min(1e-300)), float((g - r).norm() / r.norm()))
    return out


def _cotangent(shape, seed):
    """Fixed smooth upstream gradient for `preds` (keeps the comparison free of the loss's discrete anchor assignment,
    which the loss tests pin on identical predictions)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) / shape[-1] ** 0.5


@pytest.mark.parametrize("preset,res,precision", [("n", 320, "float32"), ("l", 320, "float32"), ("n", 320, "bfloat16"),
                                                  ("s", 320, "bfloat16"), ("l", 320, "bfloat16")])
def test_all_gradients_within_band_of_fp32_oracle(preset, res, precision):
    """EVERY parameter gradient of a training-mode forward/backward (fixed cotangent on preds) against the fp32 CPU
    oracle, per tensor (cosine, relative L2).  fp32 path: 2e-3.  bf16 autocast path (the MFMA kernels the bench runs):
    a randomly initialised net with two-image batch statistics amplifies bf16 rounding (the predictions already move by
    ~1e-1 max-rel), so the yardstick is the oracle itself run under CPU bf16 autocast -- the reference's own bf16 path
    -- against the same fp32 result: per tensor the HIP error may not exceed 2x the CPU-bf16 error + 0.05, and the
    median over all tensors may not exceed 1.25x the CPU-bf16 median.  Preset l = BASELINE config 4's model (C3K inside
    every C3K2, two blocks per stage)."""
    cfg = ob.PRESETS[preset]
    model = _model(seed=2, cfg=cfg).train()
    img = torch.randn(2, 3, res, res, generator=torch.Generator().manual_seed(9))
    amp = precision != "float32"
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        preds, a, s = model(img.cuda())
    assert preds.dtype == (torch.bfloat16 if amp else torch.float32)
    ct = _cotangent(tuple(preds.shape), 4)
    preds.backward(ct.to(preds.dtype).cuda())
    ps = ParamStore(2, requires_grad=True)
    p_ref, _, _ = ob.model_forward(ps, img, cfg["width"], cfg["depth"], cfg["csp"], 80, training=True)
    p_ref.backward(ct)
    e_p = _rel(preds, p_ref)
    hip = _grad_errors({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, ps)
    assert len(hip) >= (500 if preset == "l" else 240)
    worst = sorted((c, r, k) for k, (c, r) in hip.items())[:3]
    med = sorted(r for _, r in hip.values())[len(hip) // 2]
    msg = (f"\n[grad band {preset}@{res} {precision}] preds max-rel {e_p:.2e}; {len(hip)} gradient tensors: min cosine "
           f"{worst[0][0]:.5f}, median rel-L2 {med:.4f}, max rel-L2 {max(r for _, r in hip.values()):.4f}")
    if not amp:
        print(msg)
        assert e_p < 1e-3
        bad = [(k, c, r) for k, (c, r) in hip.items() if c < 0.99999 or r > 2e-3]
        assert not bad, bad[:5]
        return
    ps16 = ParamStore(2, requires_grad=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        p16, _, _ = ob.model_forward(ps16, img, cfg["width"], cfg["depth"], cfg["csp"], 80, training=True)
    p16.backward(ct.to(p16.dtype))
    cpu = _grad_errors({k: v.grad for k, v in ps16.items() if getattr(v, "grad", None) is not None}, ps)
    med16 = sorted(r for _, r in cpu.values())[len(cpu) // 2]
    print(msg + f" | CPU bf16 autocast: preds {_rel(p16, p_ref):.2e}, min cosine {min(c for c, _ in cpu.values()):.5f}, "
          f"median rel-L2 {med16:.4f}, max rel-L2 {max(r for _, r in cpu.values()):.4f}")
    assert e_p < 2 * _rel(p16, p_ref) + 1e-2
    # (a tensor that is rounding noise on the CPU under the same numeric contract too -- CPU rel-L2 > 0.5 -- is held to the wider
    # of the factor-2 band and a norm bound of 2: tests/test_gpu_fsdp.py explains)
    band = {k: (2 * cpu[k][1] + 0.05 if cpu[k][1] <= 0.5 else max(2 * cpu[k][1] + 0.05, 2.0)) for k in hip}
    worst = max(hip, key=lambda k: hip[k][1] / band[k])
    print(f"  closest to its band: {worst} at {hip[worst][1]:.3f} of {band[worst]:.3f} (CPU {cpu[worst][1]:.3f})")
    bad = [(k, hip[k], cpu[k]) for k in hip if hip[k][1] > band[k]]
    assert not bad, f"{len(bad)} tensors further from fp32 than twice the CPU bf16 path: {bad[:5]}"
    assert med <= 1.25 * med16 + 0.01, (med, med16)


def test_config5_model_half_preset_l_1280_fp16_inference_vs_fp32_oracle():
    """BASELINE config 5's model half: preset l at 1280 x 1280 in fp16 (eval mode, one image): the raw head output against
    the fp32 CPU oracle (fp16 band: 2e-2 of the tensor's max), 33600 anchors, and the inference tail (decode + NMS) on
    that output against the oracle's decode + NMS of the SAME predictions (rows exact up to fp16 rounding of the boxes)."""
    from oracle import postproc as opost
    cfg = ob.PRESETS["l"]
    model = _model(seed=4, cfg=cfg).eval()
    img = torch.randn(1, 3, 1280, 1280, generator=torch.Generator().manual_seed(21))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        preds, a, s = model(img.cuda())
    assert preds.shape == (1, 144, 33600) and preds.dtype == torch.float16 and torch.isfinite(preds.float()).all()
    ps = ParamStore(4)
    with torch.no_grad():
        p_ref, a_ref, s_ref = ob.model_forward(ps, img, cfg["width"], cfg["depth"], cfg["csp"], 80, training=False)
    err = _rel(preds, p_ref)
    print(f"\n[config 5 model half] l@1280 fp16 eval preds vs fp32 oracle: max-rel {err:.2e}")
    assert err < 2e-2
    # the inference tail on this output.  Raw class logits of a freshly initialised head sit at the -4.6 bias (nothing
    # passes a confidence threshold), so they are shifted to give thousands of candidates, as config 5's stress tensor has
    from src.hipops import ops
    from src.utils.model_utils import non_max_suppression
    shifted = preds.clone()
    shifted[:, 64:] += 5.0
    with torch.no_grad():
        y = ops.head_decode(shifted, a, s, 80)
        dets = non_max_suppression(y, conf_thres=0.25, iou_thres=0.45, nc=80)
        y_ref = opost.inference_decode(ParamStore(), shifted.float().cpu(), a.float().cpu(), s.float().cpu(), 80).half()
        want = opost.non_max_suppression(y_ref.clone(), conf_thres=0.25, iou_thres=0.45, nc=80)
    assert float((y.float().cpu() - y_ref.float()).abs().max()) <= 2.0 ** -9 * float(y_ref.float().abs().max())   # fp16 decode
    assert len(dets) == 1 and dets[0].shape == want[0].shape and dets[0].shape[0] == 300
    assert torch.equal(dets[0][:, 5].cpu().float(), want[0][:, 5].float())          # same classes in the same order


@pytest.mark.parametrize("precision", ["float16", "bfloat16"])
def test_fused_inference_equals_unfused_eval(precision):
    """Model.fuse() in 16 bit: every dense Conv block is ONE launch (bias + SiLU + residual in the conv epilogue, frozen
    weights packed once) -- against the unfused eval model (conv, BatchNorm coefficients, element-wise pass) on the same
    input, and against the fp32 oracle."""
    cfg = ob.PRESETS["s"]
    dt = getattr(torch, precision)
    img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(23))
    plain, fused = _model(seed=7, cfg=cfg).eval(), _model(seed=7, cfg=cfg).eval().fuse()
    from src.hipops import ops as hops
    stem_calls, real_stem = [], hops.stem_conv_fwd
    hops.stem_conv_fwd = lambda *a, **k: (stem_calls.append(len(a) > 6 and a[5] is not None and a[6] == 1), real_stem(*a, **k))[1]
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=dt):
            p0 = plain(img.cuda())[0]
            p1 = fused(img.cuda())[0]
            p2 = fused(img.cuda())[0]                      # second call: the cached packed weights
    finally:
        hops.stem_conv_fwd = real_stem
    from src.model.model_blocks import C3K
    c3k = [m for m in fused.modules() if type(m) is C3K]
    assert c3k and all(m.__dict__.get("_pair") is not None for m in c3k), "C3K's two entry convs must run as one stacked conv"
    assert stem_calls[-2:] == [True, True], "the fused model's stem must take the stem kernel with bias + SiLU in its epilogue"
    ps = ParamStore(7)
    with torch.no_grad():
        p_ref, _, _ = ob.model_forward(ps, img, cfg["width"], cfg["depth"], cfg["csp"], 80, training=False)
    e01, e1r, e0r = _rel(p1, p0), _rel(p1, p_ref), _rel(p0, p_ref)
    print(f"\n[fused inference {precision}] fused vs unfused {e01:.2e}; vs fp32 oracle: fused {e1r:.2e}, unfused {e0r:.2e}")
    assert torch.equal(p1, p2)
    tol = 2e-2 if precision == "float16" else 8e-2
    assert e1r < tol and e1r < 2 * e0r + 1e-3


@pytest.mark.parametrize("fused", [False, True])
def test_replayed_inference_equals_eager_inference(fused):
    """Model.inference replays forward + decode as one hipGraph per input shape (src/model/infer_graph.py): detections
    bit-identical to the launch-by-launch path on every image, a new shape captures a new graph, in-place weight updates
    are seen (unfused: the kernels read the live parameters; a changed version counter re-captures), .half() drops the
    graphs, and a training-mode or grad-enabled call never replays."""
    from src.model.model_builder import Model
    g = torch.Generator().manual_seed(31)
    imgs = [torch.randn(1, 3, 320, 320, generator=g).cuda() for _ in range(3)] + [torch.randn(2, 3, 256, 384, generator=g).cuda()]
    model = _model(seed=9, cfg=ob.PRESETS["s"]).eval()
    with torch.no_grad():
        for lvl in model.head.cls:                   # enough confident anchors for the NMS to have work
            lvl[-1].weight.mul_(0.05)
            lvl[-1].bias.copy_(torch.linspace(-2.0, 1.0, 80, device="cuda"))
    if fused:
        model.fuse()

    def both(img, conf=0.05):
        Model.graph_inference = False
        try:
            eager = model.inference(img, conf_thres=conf)
        finally:
            Model.graph_inference = True
        return eager, model.inference(img, conf_thres=conf)

    with torch.autocast("cuda", dtype=torch.float16):
        for img in imgs + imgs[:1]:
            eager, replayed = both(img)
            assert len(eager) == len(replayed) == img.shape[0] and sum(e.shape[0] for e in eager) > 0
            for e, r in zip(eager, replayed):
                assert torch.equal(e, r)
        graphs = model._infer_graphs
        assert graphs is not None and graphs.disabled is None and len(graphs.entries) == 2
        first = replayed = model.inference(imgs[0], conf_thres=0.05)
        # in-place update of a weight: the next call must see it
        with torch.no_grad():
            w = model.head.box[0][-1].weight
            w.mul_(1.5)
        eager, replayed = both(imgs[0])
        assert all(torch.equal(e, r) for e, r in zip(eager, replayed))
        assert not all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(first, replayed)), "stale weights replayed"
    model.half()
    assert model._infer_graphs is None
    eager, replayed = both(imgs[0].half())
    assert all(torch.equal(e, r) for e, r in zip(eager, replayed))
    with torch.enable_grad():
        assert model._infer_graphs.run(model, imgs[0].half(), 0.05, 0.45) is None
