"""-m gpu: HipAdamW (one-launch AdamW) against torch.optim.AdamW -- eager, GradScaler protocol, state dict
round trip, and inside a captured graph with a learning-rate change between replays."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (64,), (7,), (128, 64, 1, 1), (1,), (5000,)]
    return [torch.randn(*s, generator=g).to(dtype).cuda().requires_grad_(True) for s in shapes]


def _set_grads(ps, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    for p in ps:
        p.grad = (torch.randn(*p.shape, generator=g) * scale).to(p.dtype).cuda()


def test_matches_torch_adamw_over_steps_and_lr_changes():
    from src.training.fused_adamw import HipAdamW
    a, b = _params(0), _params(0)
    oa = HipAdamW(a, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ob = torch.optim.AdamW(b, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for s in range(6):
        if s == 3:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 2.5e-4          # what ReduceLROnPlateau does
        _set_grads(a, 100 + s), _set_grads(b, 100 + s)
        oa.step(), ob.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-7), float((x - y).abs().max())
    sa, sb = oa.state[a[0]], ob.state[b[0]]
    assert float(sa["step"]) == 6 == float(sb["step"])
    assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-7)       # torch uses lerp: last-bit differences
    assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-9)


def test_grad_scaler_protocol_and_state_dict_round_trip():
    from src.training.fused_adamw import HipAdamW
    a, b = _params(1), _params(1)
    oa = HipAdamW(a, lr=1e-3, weight_decay=1e-2)
    ob = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-2)
    scale = torch.tensor([1024.0], device="cuda")
    # step 1: finite scaled gradients -> unscaled update; step 2: overflow flagged -> skipped entirely
    _set_grads(a, 7, 1024.0), _set_grads(b, 7, 1.0)
    oa.grad_scale, oa.found_inf = scale, torch.zeros(1, device="cuda")
    oa.step(), ob.step()
    before = [p.detach().clone() for p in a]
    _set_grads(a, 8, 1024.0)
    oa.found_inf = torch.ones(1, device="cuda")
    oa.step()
    del oa.grad_scale, oa.found_inf          # what GradScaler.step does after the call
    for x, y, z in zip(a, b, before):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6) and torch.equal(x, z)
    assert float(oa.state[a[0]]["step"]) == 1
    # state dict -> fresh optimizer continues identically
    import copy
    sd = copy.deepcopy(oa.state_dict())        # what a checkpoint round trip gives (a live state_dict aliases the tensors)
    c = [p.detach().clone().requires_grad_(True) for p in a]
    oc = HipAdamW(c, lr=1e-3, weight_decay=1e-2)
    oc.load_state_dict(sd)
    _set_grads(a, 9), _set_grads(c, 9)
    oa.step(), oc.step()
    for x, y in zip(a, c):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)
    assert float(oc.state[c[0]]["step"]) == 2


def test_inside_a_captured_graph_with_lr_change_between_replays():
    from src.training.fused_adamw import HipAdamW
    a, b = _params(2), _params(2)
    oa = HipAdamW(a, lr=1e-3, weight_decay=0.0)
    ob = torch.optim.AdamW(b, lr=1e-3, weight_decay=0.0)
    static = [torch.zeros_like(p) for p in a]

    def produce(ps):                       # gradient = a fresh tensor every time (like autograd), values from `static`
        for p, s in zip(ps, static):
            p.grad = s * 1.0

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s_ in static:
            s_.normal_()
        produce(a), produce(b)
        oa.step(), ob.step()               # eager warm-up step (allocates tables / state)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for p in a:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        produce(a)
        oa.step()
    oa.finish_capture()
    gen = torch.Generator(device="cuda").manual_seed(3)
    for r in range(4):
        if r == 2:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 1e-4
            oa.sync_hyper()
        for s_ in static:
            s_.normal_(generator=gen)
        produce(b)
        g.replay(), ob.step()
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=5e-6, atol=5e-7), float((x - y).abs().max())
    assert float(oa.state[a[0]]["step"]) == 5


def test_device_grad_scaler_follows_torch_gradscaler_step_for_step():
    """DeviceGradScaler + HipAdamW (found_inf, skipped / unscaled update and the scale's growth / backoff all on the device,
    four launches: csrc/optim.hip yolo_adamw_amp_step) against torch.amp.GradScaler + torch.optim.AdamW on the same scaled
    gradients: finite, overflow (inf), finite, nan, finite, finite with growth_interval 2 -- parameters and the scale
    after every step (reference protocol: src/training/train_model.py:195-208,247-253)."""
    from src.training.fused_adamw import DeviceGradScaler, HipAdamW
    a, b = _params(3), _params(3)
    oa = HipAdamW(a, lr=1e-3, weight_decay=1e-2)
    ob = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-2)
    mine = DeviceGradScaler("cuda", init_scale=4096.0, growth_interval=2)
    oa.device_amp = mine
    ref = torch.amp.GradScaler("cuda", init_scale=4096.0, growth_interval=2)
    ref.scale(torch.zeros(1, device="cuda"))                       # lazy init of torch's scale tensor
    poison = {1: float("inf"), 3: float("nan")}
    skipped = []
    for s in range(6):
        sc = ref.get_scale()
        assert mine.get_scale() == sc, (s, mine.get_scale(), sc)
        _set_grads(a, 40 + s, sc), _set_grads(b, 40 + s, sc)        # gradients of the scaled loss
        if s in poison:
            a[2].grad[3] = poison[s]
            b[2].grad[3] = poison[s]
        oa.step()
        ref.step(ob)
        ref.update()
        skipped.append(mine.last_step_skipped())
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-6), (s, float((x - y).abs().max()))
    assert skipped == [False, True, False, True, False, False]
    assert mine.get_scale() == ref.get_scale() == 2048.0           # 4096 -> 2048 (inf) -> 1024 (nan) -> 2048 (two good steps)
    assert float(oa.state[a[0]]["step"]) == 4 == float(ob.state[b[0]]["step"])
    sd = mine.state_dict()
    again = DeviceGradScaler("cuda")
    again.load_state_dict(sd)
    assert again.get_scale() == 2048.0 and int(again.tracker) == int(mine.tracker)


def test_loss_kernel_applies_the_scale_in_fp32_before_rounding():
    """fp16 predictions: gradients of ~1e-6 are subnormal in fp16; scaled by 65536 INSIDE the kernel they keep fp32's
    precision (scaling the rounded fp16 gradient afterwards, as `scale * loss` through a post-pass did, loses them)."""
    from src.hipops import ops
    from src.model.losses import PackedTargets
    g = torch.Generator().manual_seed(5)
    n, nc, a = 8, 80, 8400
    preds = (torch.randn(n, 64 + nc, a, generator=g) * 0.5).half().cuda()
    from src.utils.model_utils import make_anchors_cached
    anchors, strides = make_anchors_cached(((80, 80), (40, 40), (20, 20)), (8.0, 16.0, 32.0), torch.float16, torch.device("cuda", 0))
    gts = [torch.tensor([[100., 120., 60., 40., 3.]]).cuda(), torch.tensor([[200., 80., 30., 90., 7.], [50., 50., 20., 20., 1.]]).cuda()] * 4
    pk = PackedTargets(gts, "cuda")
    scale = torch.tensor([65536.0], device="cuda")
    out1, d1, _ = ops.loss_fwd_bwd(preds, anchors, strides, *pk.as_tuple(), nc, 1.5, 1.0, True)
    out2, d2, _ = ops.loss_fwd_bwd(preds, anchors, strides, *pk.as_tuple(), nc, 1.5, 1.0, True, scale)
    out3, d3, _ = ops.loss_fwd_bwd(preds.float(), anchors.float(), strides.float(), *pk.as_tuple(), nc, 1.5, 1.0, True)
    assert torch.equal(out1, out2)                                  # the loss VALUE is never scaled
    want = d3 * 65536.0
    small = (d3.abs() < 6e-5) & (d3.abs() > 2e-7)                   # unscaled values in fp16's subnormal range
    assert int(small.sum()) > 1000
    rel_in = float(((d2.float() - want).abs() / want.abs())[small].mean())
    rel_post = float(((d1.float() * 65536.0 - want).abs() / want.abs())[small].mean())
    print(f"\n[fp16 loss scaling] mean relative error of the {int(small.sum())} gradients that are subnormal before scaling: "
          f"scaled in the kernel (fp32) {rel_in:.2e}, scaled after the fp16 rounding {rel_post:.2e}")
    assert rel_in < 1e-3 and torch.isfinite(d2.float()).all() and rel_post > 4 * rel_in
