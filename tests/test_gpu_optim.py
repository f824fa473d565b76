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
