"""-m gpu: the reference's call sequence end to end on one GPU -- init_distributed_mode (RCCL, world 1),
prepare_{ddp,fsdp2}_model, get_optimizer, get_data_loaders (synthetic), train() for one short epoch incl.
validation decode/metrics and the rank-0 checkpoint -- i.e. what scripts/distributed_training.py does."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])


@pytest.fixture(scope="module")
def pg():
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    from src.training.distributed_setup import cleanup_distribute_mode, init_distributed_mode
    yield init_distributed_mode("cuda")
    cleanup_distribute_mode()


@pytest.mark.parametrize("mode,precision,captured", [("ddp", "bfloat16", False), ("ddp", "float32", False),
                                                     ("ddp", "float16", False), ("fsdp2", "bfloat16", False),
                                                     ("fsdp2", "float16", False),
                                                     ("fsdp", "bfloat16", False), ("fsdp", "float32", False),
                                                     ("ddp", "bfloat16", True), ("ddp", "float32", True), ("ddp", "float16", True),
                                                     ("fsdp2-native", "bfloat16", None), ("fsdp2-native", "float32", None)])
def test_train_one_epoch_like_the_script(pg, mode, precision, captured, tmp_path):
    from src.data.data_loader import get_data_loaders
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.train_model import train
    from src.training.utils_train import get_optimizer, prepare_ddp_model, prepare_fsdp2_model, prepare_fsdp_model
    rank, world, gpu = pg
    torch.manual_seed(0)
    model = Model(**NANO, num_classes=80)
    native = mode == "fsdp2-native"          # config key training.fsdp2.native_shard: the captured sharded step (sharded_step.py)
    mode = "fsdp2" if native else mode
    wrap = {"ddp": prepare_ddp_model, "fsdp2": prepare_fsdp2_model, "fsdp": prepare_fsdp_model}[mode]
    # captured=True: the default drop-in path -- the DDP-wrapped model stepped through TrainStepRunner on its .module
    model = wrap(model=model, device_id=gpu, config={"precision": precision, "find_unused_parameters": False,
                                                      "sharding_strategy": "FULL_SHARD", "auto_wrap_policy_min_params": 20000,
                                                      "native_shard": native},
                 world_size=world, device="cuda")
    tr, va = get_data_loaders("/nonexistent/train", "/nonexistent/val", "", "", batch_size=4, is_test=True, device="cuda",
                              num_classes=80, res=160)
    opt, sched = get_optimizer(model, lr=1e-4, weight_decay=1e-4, patience=3, factor=0.5)
    from src.training.fused_adamw import HipAdamW
    # FSDP1's parameters are views of its flat shards, re-pointed every step: torch.optim.AdamW; everything else (plain
    # parameters, FSDP2's DTensor shards) steps in one launch
    assert type(opt) is (torch.optim.AdamW if mode == "fsdp" else HipAdamW)
    masters = (lambda: [model._native_shard["state"].master.detach().float().clone()]) if native else None
    before = masters() if native else [p.detach().float().clone() for p in model.parameters()][:3]
    train(model=model, train_loader=tr, val_loader=va, optimizer=opt, scheduler=sched,
          criterion=YoloDFLQFLoss(num_classes=80), initial_epoch=0, num_epochs=1, device=gpu, num_classes=80, rank=rank,
          checkpoint_dir=str(tmp_path), distributed_mode=mode, precision=precision, conf_threshold=0.01,
          captured_step=captured)
    if captured and precision == "float16":             # the captured route scales on the device: GradScaler's rules, no GradScaler
        amp = getattr(opt, "device_amp", None)
        assert amp is not None and amp.get_scale() in (65536.0, 32768.0, 16384.0, 8192.0) and torch.isfinite(amp.state).all()
    after = masters() if native else [p.detach().float() for p in model.parameters()][:3]
    assert any(not torch.equal(a, b) for a, b in zip(before, after)), "parameters did not move"
    assert all(torch.isfinite(a).all() for a in after)
    ck = torch.load(os.path.join(str(tmp_path), "model_epoch_1.pth"), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and "model_state" in ck and "optimizer_state" in ck
    if native:      # full fp32 tensors under canonical names + a torch.optim.AdamW-shaped optimizer state: a bare Model resumes from it
        bare = Model(**NANO, num_classes=80)
        bare.load_state_dict(ck["model_state"])
        o2 = torch.optim.AdamW([p for p in bare.parameters() if p.requires_grad], lr=1e-4)
        o2.load_state_dict(ck["optimizer_state"])
        assert all(v.dtype == torch.float32 for v in ck["model_state"].values() if v.is_floating_point())
        # ... and the sharded model itself resumes (masters, compute copies and moments from the full tensors)
        from src.training.utils_train import load_checkpoint
        assert load_checkpoint(model, opt, os.path.join(str(tmp_path), "model_epoch_1.pth")) == 1
        assert torch.equal(masters()[0], after[0])
    keys = set(ck["model_state"].keys())
    assert any(k.endswith("net.p1.0.conv.weight") for k in keys) and any(k.endswith("head.dfl.conv.weight") for k in keys)


def test_two_graph_data_parallel_step_equals_single_graph_step(pg):
    """The N > 1 step of TrainStepRunner (graph A: fwd + backward of head and neck + pack bucket A, RCCL all-reduce beside
    graph B: backward of the backbone + pack bucket B, second all-reduce, graph C: unpack + AdamW) on a one-rank RCCL
    group must leave the gradients of the single-graph step, with two buckets or one; with bf16-compressed buckets it
    must stay within bf16 rounding of them."""
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.graph_step import TrainStepRunner
    g = torch.Generator().manual_seed(11)
    img = torch.randn(2, 3, 160, 160, generator=g).cuda()
    gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                      torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]

    def run(force, comm_dtype, buckets=2):
        torch.manual_seed(0)
        model = Model(**NANO, num_classes=80).cuda().train()
        # lr 0: the weights stay put, so the gradients of the last step are comparable across runs (an Adam update is
        # sign-like for near-zero gradients and would amplify the last-bit noise of the atomically summed statistics)
        opt = torch.optim.AdamW(model.parameters(), lr=0.0, weight_decay=0.0, capturable=True, fused=True)
        # fp32 compute: the only run-to-run noise left is the last bits of the atomically summed batch statistics
        r = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, "float32", use_graph=True,
                            grad_comm_dtype=comm_dtype, force_comm=force, buckets=buckets)
        r.capture(img, PackedTargets(gts, img.device), warmup=1)
        for _ in range(2):
            r.step()
        torch.cuda.synchronize()
        assert (r.graph2 is not None) == force and (r.graph_b is not None) == (force and buckets == 2)
        st = opt.state[next(iter(model.parameters()))]
        assert int(st["step"]) >= 2                                     # the optimizer really stepped (in graph 2)
        return [p.grad.detach().float().clone() for p in model.parameters() if p.grad is not None]

    # one graph (no exchange) / staged backward with two buckets all-reduced beside the backbone's backward / one flat
    # bucket / two bf16-compressed buckets
    base, same, flat1, comp = run(False, None), run(True, None), run(True, None, buckets=1), run(True, torch.bfloat16)
    gmax = max(float(a.abs().max()) for a in base)
    for a, b, f, c in zip(base, same, flat1, comp):
        scale = a.abs().max().clamp_min(1e-3 * gmax)       # some gradients are mathematically zero (noise only)
        assert torch.isfinite(b).all() and torch.isfinite(c).all() and torch.isfinite(f).all()
        assert (a - b).abs().max() / scale < 1e-3, "staged backward / pack / all-reduce / unpack changed the gradients"
        assert (a - f).abs().max() / scale < 1e-3, "single-bucket exchange changed the gradients"
        assert (a - c).abs().max() / scale < 1e-2, "bf16-compressed exchange outside bf16 rounding"
    assert any(not torch.equal(a, c) for a, c in zip(base, comp))


@pytest.mark.parametrize("mode", ["ddp", "fsdp2"])
def test_checkpoint_round_trip_across_wrappers(pg, mode, tmp_path):
    """A checkpoint written from a DDP- or FSDP2-wrapped model (full tensors, gathered by a collective for the sharded
    case) loads into a bare Model with identical inference output, and resumes a freshly wrapped model + optimizer."""
    from src.model.model_builder import Model
    from src.training.utils_train import (checkpoint_states, get_optimizer, load_checkpoint, prepare_ddp_model,
                                          prepare_fsdp2_model, save_checkpoint)
    rank, world, gpu = pg
    wrap = prepare_ddp_model if mode == "ddp" else prepare_fsdp2_model
    conf = {"precision": "float32", "find_unused_parameters": False}
    torch.manual_seed(5)
    model = wrap(model=Model(**NANO, num_classes=80), device_id=gpu, config=conf, world_size=world, device="cuda")
    opt, _ = get_optimizer(model, lr=1e-3, weight_decay=1e-4, patience=3, factor=0.5)
    img = torch.randn(2, 3, 160, 160, device="cuda")
    model.train()
    preds, _, _ = model(img)
    preds.float().square().mean().backward()
    opt.step()
    save_checkpoint(model, opt, 4, 0.25, checkpoint_dir=str(tmp_path), states=checkpoint_states(model, opt))
    path = os.path.join(str(tmp_path), "model_epoch_4.pth")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert all(type(v) is torch.Tensor for v in ck["model_state"].values()), "checkpoint holds sharded / wrapped tensors"
    model.eval()
    with torch.no_grad():
        want = model(img)[0].float()
    bare = Model(**NANO, num_classes=80).cuda()
    bare.load_weights(path)
    bare.eval()
    with torch.no_grad():
        got = bare(img)[0].float()
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), float((got - want).abs().max())
    fresh = wrap(model=Model(**NANO, num_classes=80), device_id=gpu, config=conf, world_size=world, device="cuda")
    opt2, _ = get_optimizer(fresh, lr=1e-3, weight_decay=1e-4, patience=3, factor=0.5)
    assert load_checkpoint(fresh, opt2, path, map_location="cuda") == 4
    fresh.eval()
    with torch.no_grad():
        again = fresh(img)[0].float()
    assert torch.allclose(again, want, rtol=1e-4, atol=1e-5)
    st = opt2.state_dict()["state"]
    assert len(st) > 0 and all("exp_avg" in v for v in st.values())


@pytest.mark.parametrize("group,precision", [(1, "float32"), (3, "float32"), (8, "float32"), (3, "bfloat16"), (8, "bfloat16")])
def test_runner_gradients_equal_plain_autograd(group, precision, monkeypatch):
    """TrainStepRunner runs the conv weight gradients on a side stream that is joined lazily, several layers per
    sync point; the gradients it leaves in .grad -- eagerly and after graph replays -- must be those of a plain
    `loss.backward()`.  (Regression: a gradient tensor that was also referenced by the pending-work queue was CLONED
    by autograd's AccumulateGrad before the side stream had written it: stale gradients, training still converged.)"""
    from src.hipops import functions as F_
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.graph_step import TrainStepRunner
    monkeypatch.setattr(F_, "WGRAD_GROUP", group)
    g = torch.Generator().manual_seed(12)
    img = torch.randn(2, 3, 160, 160, generator=g).cuda()
    gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                      torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
    packed = PackedTargets(gts, img.device)
    torch.manual_seed(0)
    model = Model(**NANO, num_classes=80).cuda().train()
    crit = YoloDFLQFLoss(num_classes=80)
    if precision == "bfloat16":
        # bf16 runs in DETERMINISTIC mode (functions.deterministic_stats: every BatchNorm statistic through the fixed-order
        # two-level reduction instead of float atomics): the runner and plain autograd then launch the same kernels on the
        # same values, so the gradients must be BIT-IDENTICAL -- with the real detection loss, whose nearest-centre
        # assignment turns any stale or reordered value into an O(1) difference.
        monkeypatch.setattr(F_, "DETERMINISTIC", True)

    def grads():
        return [p.grad.detach().float().clone() for p in model.parameters() if p.grad is not None]

    amp = torch.bfloat16 if precision == "bfloat16" else None
    tol = 1e-3              # fp32: the VALU weight-gradient kernel sums row slabs with float atomics (last-bit noise)

    def plain(image):                                       # plain autograd: per-layer fork/join inside each backward
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            preds, anchors, strides = model(image)
            loss = crit(preds, packed, anchors, strides)[0]
        loss.backward()
        torch.cuda.synchronize()
        return grads()

    def check(want, got, what):
        gmax = max(float(a.abs().max()) for a in want)
        for i, (a, b) in enumerate(zip(want, got)):
            if amp is None:
                scale = a.abs().max().clamp_min(1e-3 * gmax)
                assert (a - b).abs().max() / scale < tol, (what, float((a - b).abs().max() / scale))
            else:
                assert torch.equal(a, b), (what, i, float((a - b).abs().max()), float(a.abs().max()))

    want = plain(img)
    if amp is not None:
        check(want, plain(img), "plain autograd twice (deterministic mode)")
    plain(torch.randn(2, 3, 160, 160, generator=g).cuda())  # freed blocks now hold ANOTHER batch's gradients
    opt = torch.optim.AdamW(model.parameters(), lr=0.0, weight_decay=0.0, capturable=True, fused=True)
    model.zero_grad(set_to_none=True)
    r = TrainStepRunner(model, crit, opt, precision, use_graph=False)
    r._fwd_bwd(img, packed)
    torch.cuda.synchronize()
    check(want, grads(), "eager runner step")
    r._fwd_bwd(img, packed)                                 # .grad present: autograd accumulates at once, so the runner
    torch.cuda.synchronize()                                # must not defer the weight gradients of this call
    check([2 * a for a in want], grads(), "accumulating second backward")
    r = TrainStepRunner(model, crit, opt, precision, use_graph=True)
    r.capture(img, packed, warmup=1)
    with torch.no_grad():                                   # new weights: last replay's gradients are now WRONG ones,
        for p in model.parameters():                        # and the static gradient buffers are poisoned
            p.mul_(1.02)
            if p.grad is not None:
                p.grad.fill_(1e6)
    r.step()
    torch.cuda.synchronize()
    got = grads()
    check(plain(img), got, "graph replay")


def test_ddp_wrapper_built_on_the_default_stream_is_stepped_eagerly_not_captured(pg, tmp_path):
    """A DistributedDataParallel constructed the plain way (default stream) keeps AccumulateGrad nodes alive that run on the
    default stream; capturing the step with them killed the process inside hipStreamEndCapture (round 2,
    profiles/r2_capture_probe.log).  TrainStepRunner.capture now MEASURES where the nodes run on its warm-up step and keeps
    stepping eagerly: train(captured_step=True) completes, nothing is captured, the weights move."""
    import warnings
    from torch.nn.parallel import DistributedDataParallel as DDP
    from src.data.data_loader import get_data_loaders
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training import train_model as tm
    from src.training.utils_train import get_optimizer
    rank, world, gpu = pg
    torch.manual_seed(0)
    model = DDP(Model(**NANO, num_classes=80).cuda(), device_ids=[gpu])          # NOT prepare_ddp_model: default stream
    img = torch.randn(2, 3, 160, 160, device="cuda")
    model(img)[0].float().mean().backward()         # the reducer's nodes have run once on the default stream
    model.zero_grad(set_to_none=True)
    tr, va = get_data_loaders("/nonexistent/train", "/nonexistent/val", "", "", batch_size=4, is_test=True, device="cuda",
                              num_classes=80, res=160)
    opt, sched = get_optimizer(model, lr=1e-4, weight_decay=1e-4, patience=3, factor=0.5)
    runners = []
    orig = tm.CapturedTraining.step

    def spy(self, images, boxes):
        out = orig(self, images, boxes)
        runners.append(self.runner)
        return out
    tm.CapturedTraining.step = spy
    before = [p.detach().clone() for p in model.parameters()][:3]
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            tm.train(model=model, train_loader=tr, val_loader=va, optimizer=opt, scheduler=sched,
                     criterion=YoloDFLQFLoss(num_classes=80), initial_epoch=0, num_epochs=1, device=gpu, num_classes=80, rank=rank,
                     checkpoint_dir=str(tmp_path), distributed_mode="ddp", precision="bfloat16", conf_threshold=0.01,
                     captured_step=True)
    finally:
        tm.CapturedTraining.step = orig
    r = runners[-1]
    assert r is not None and r.graph is None and r.capture_refused, "the step was captured with foreign AccumulateGrad nodes alive"
    assert any("NOT captured" in str(x.message) for x in w)
    after = [p.detach() for p in model.parameters()][:3]
    assert any(not torch.equal(a, b) for a, b in zip(before, after)) and all(torch.isfinite(a).all() for a in after)


def test_deterministic_mode_two_runs_of_the_captured_step_are_bit_identical(monkeypatch):
    """Deterministic mode (torch.use_deterministic_algorithms(True) / functions.DETERMINISTIC): two independent runs of the
    same captured bf16 training -- model, capture, three replays with the optimizer in the graph -- end with bit-identical
    weights, BatchNorm buffers and loss scalars; without the mode the float-atomic statistics make them differ."""
    from src.hipops import functions as F_
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.fused_adamw import HipAdamW
    from src.training.graph_step import TrainStepRunner
    g = torch.Generator().manual_seed(31)
    img = torch.randn(2, 3, 160, 160, generator=g).cuda()
    gts = [torch.cat([torch.rand(4, 2, generator=g) * 160, torch.rand(4, 2, generator=g) * 60 + 8,
                      torch.randint(0, 80, (4, 1), generator=g).float()], 1).cuda() for _ in range(2)]

    def run():
        torch.manual_seed(0)
        model = Model(**NANO, num_classes=80).cuda().train()
        opt = HipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
        r = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, "bfloat16", use_graph=True)
        r.capture(img, PackedTargets(gts, img.device), warmup=1)
        for _ in range(3):
            r.step()
        torch.cuda.synchronize()
        assert r.graph is not None and r.opt_in_graph
        return [t.detach().clone() for t in list(model.parameters()) + [b for b in model.buffers() if b.is_floating_point()]] + [r.scalars.clone()]

    assert torch.are_deterministic_algorithms_enabled() is False
    torch.use_deterministic_algorithms(True, warn_only=True)       # the documented switch; F_.DETERMINISTIC is the package's own
    try:
        assert F_.deterministic_stats()
        a, b = run(), run()
    finally:
        torch.use_deterministic_algorithms(False)
    assert not F_.deterministic_stats()
    bad = [i for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
    assert not bad, (len(bad), len(a))
    c, d = run(), run()                                             # default mode: float atomics, run-to-run noise
    differs = sum(1 for x, y in zip(c, d) if not torch.equal(x, y))
    print(f"\n[deterministic mode] {len(a)} tensors bit-identical across two runs; default mode: {differs} of {len(c)} differ")


def test_captured_step_on_new_batches_equals_eager_steps():
    """TrainStepRunner.step_batch refills the static image / target buffers (different images, different numbers of
    boxes per image, an image without boxes, fewer boxes than at capture) and replays: after three such steps the
    weights equal those of a twin model stepped eagerly on the same batches (fp32, plain AdamW semantics)."""
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.fused_adamw import HipAdamW
    from src.training.graph_step import TrainStepRunner
    g = torch.Generator().manual_seed(21)

    def batch(counts):
        img = torch.randn(len(counts), 3, 160, 160, generator=g).cuda()
        gts = [torch.cat([torch.rand(c, 2, generator=g) * 160, torch.rand(c, 2, generator=g) * 60 + 8,
                          torch.randint(0, 80, (c, 1), generator=g).float()], 1) for c in counts]
        return img, gts

    batches = [batch([3, 5]), batch([1, 0]), batch([7, 2]), batch([2, 2])]
    torch.manual_seed(0)
    a = Model(**NANO, num_classes=80).cuda().train()
    b = Model(**NANO, num_classes=80).cuda().train()
    b.load_state_dict(a.state_dict())
    start = [p.detach().clone() for p in a.parameters()]
    crit = YoloDFLQFLoss(num_classes=80)
    oa, ob = HipAdamW(a.parameters(), lr=1e-4, weight_decay=1e-2), HipAdamW(b.parameters(), lr=1e-4, weight_decay=1e-2)
    ra = TrainStepRunner(a, crit, oa, "float32", use_graph=True)
    ra.capture_for_batches(*batches[0], boxes_per_image=8, warmup=1)        # capture steps on batch 0 (warm-up + none)
    rb = TrainStepRunner(b, crit, ob, "float32", use_graph=False)
    for _ in range(1):                                                      # the same eager warm-up steps on the twin
        rb._eager_step(batches[0][0], [t.cuda() for t in batches[0][1]])
    losses = []
    for img, gts in batches[1:]:
        la = ra.step_batch(img, gts)
        assert la is not None
        lb, _ = rb._eager_step(img, [t.cuda() for t in gts])
        losses.append((float(la), float(lb)))
    torch.cuda.synchronize()
    # the first new batch sees identical weights: sharp; later ones see weights that Adam's sign-like update has moved
    # apart by last-bit gradient noise (and the loss's anchor assignment is discrete)
    for k, (x, y) in enumerate(losses):
        assert abs(x - y) <= (1e-4 if k == 0 else 5e-3) * abs(y) + 1e-5, losses
    # Adam's update is sign-like (lr * m / sqrt(v)): last-bit noise of the atomically summed statistics moves a weight
    # with a near-zero gradient by up to ~lr per step, so the weights are compared at a few lr; the per-step losses
    # above, which see the updated weights, are the sharp check
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p - q).abs().max()) <= 4e-4, (n, float((p - q).abs().max()))
    moved = max(float((p - q0).abs().max()) for p, q0 in zip(a.parameters(), start))
    assert moved > 2e-4, "the captured steps did not update the weights"
    big = batch([9, 9])
    assert ra.step_batch(*big) is None                                      # 18 boxes > capacity 16: caller falls back
    assert ra.step_batch(torch.randn(2, 3, 128, 128).cuda(), batches[1][1]) is None
    # the fallback is an EAGER optimizer step (new gradient tensors: HipAdamW rebuilds the job table the captured
    # launch reads); restore_capture() must put the captured table back before the next replay
    ra._eager_step(big[0], [t.cuda() for t in big[1]])
    rb._eager_step(big[0], [t.cuda() for t in big[1]])
    oa.restore_capture()
    last = batch([4, 1])
    la = ra.step_batch(*last)
    lb, _ = rb._eager_step(last[0], [t.cuda() for t in last[1]])
    torch.cuda.synchronize()
    assert la is not None and abs(float(la) - float(lb)) <= 5e-3 * abs(float(lb)) + 1e-5, (float(la), float(lb))
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.isfinite(p).all() and float((p - q).abs().max()) <= 8e-4, (n, float((p - q).abs().max()))


def test_graph_replays_reproduce_the_eager_forward():
    """A captured forward must give the eager result on EVERY replay.  (Regression: the statistics accumulators were
    zeroed by hipMemsetAsync, which a capture turns into a memset node; replays did not always order it before the
    kernels accumulating into the buffer, so BatchNorm went wrong from the second replay on.  Everything the step
    zeroes now goes through a zero-fill kernel.)"""
    from src.model.model_builder import Model
    torch.manual_seed(0)
    model = Model(**NANO, num_classes=80).cuda().train()
    model.prepack = True
    img = torch.randn(2, 3, 160, 160, generator=torch.Generator().manual_seed(5)).cuda()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            ref = model(img)[0].float().clone()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        out = model(img)[0]
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        assert err < 1e-4, f"replay {i}: {err}"


def test_capture_survives_pinned_memory_owners_dropped_with_the_collector_enabled(pg):
    """Regression for the run-order dependent abort of round 1 (commit 3e4e500): objects that own pinned host memory --
    an optimizer's job table, the StaticTargets of an earlier runner, gradient-bucket tables -- dropped in reference
    cycles right before a capture, with the garbage collector ENABLED and primed to run.  TrainStepRunner.capture
    collects before the capture and keeps the collector off until all graphs are captured."""
    import gc
    from src.model.losses import PackedTargets, StaticTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.fused_adamw import HipAdamW
    from src.training.graph_step import TrainStepRunner
    g = torch.Generator().manual_seed(31)
    img = torch.randn(2, 3, 160, 160, generator=g).cuda()
    gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                      torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
    torch.manual_seed(0)
    model = Model(**NANO, num_classes=80).cuda().train()
    crit = YoloDFLQFLoss(num_classes=80)
    assert gc.isenabled()

    class Cycle:                                   # only the collector can free what hangs off a reference cycle
        def __init__(self, *owned):
            self.owned, self.me = owned, self
    # an earlier runner with its optimizer, static targets and bucket tables: stepped, then dropped in cycles
    old_opt = HipAdamW(model.parameters(), lr=1e-5)
    old = TrainStepRunner(model, crit, old_opt, "float32", use_graph=True, force_comm=True)
    old.capture_for_batches(img, [t.cpu() for t in gts], boxes_per_image=8, warmup=1)
    old.step()
    torch.cuda.synchronize()
    Cycle(old, old_opt, StaticTargets(2, 64, img.device))
    del old, old_opt
    gc.set_threshold(1, 1, 1)                      # the collector now runs at (almost) every allocation
    try:
        opt = HipAdamW(model.parameters(), lr=1e-5)
        r = TrainStepRunner(model, crit, opt, "float32", use_graph=True, force_comm=True)
        r.capture(img, PackedTargets(gts, img.device), warmup=1)
        for _ in range(2):
            loss = r.step()
        torch.cuda.synchronize()
    finally:
        gc.set_threshold(700, 10, 10)
    assert gc.isenabled() and torch.isfinite(loss)


def test_eval_after_captured_steps_uses_the_updated_weights():
    """The one-launch weight packing serves packed copies keyed by the parameter's version counter, which the raw AdamW
    kernel does not bump: an evaluation pass in the same process must not read the copies packed BEFORE the last
    optimizer step.  Captured steps, then model.eval(): its output must equal that of a fresh model loaded from the
    state dict."""
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.fused_adamw import HipAdamW
    from src.training.graph_step import TrainStepRunner
    g = torch.Generator().manual_seed(41)
    img = torch.randn(2, 3, 160, 160, generator=g).cuda()
    gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                      torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
    torch.manual_seed(0)
    model = Model(**NANO, num_classes=80).cuda().train()
    opt = HipAdamW(model.parameters(), lr=3e-3)                  # large steps: stale weights would show
    r = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, "bfloat16", use_graph=True)
    r.capture(img, PackedTargets(gts, img.device), warmup=1)
    for _ in range(3):
        r.step()
    torch.cuda.synchronize()
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        got = model(img)[0].float()
    twin = Model(**NANO, num_classes=80).cuda()
    twin.load_state_dict(model.state_dict())
    twin.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        want = twin(img)[0].float()
    assert torch.equal(got, want), float((got - want).abs().max())


def _ddp_rank(rank, world, port, out):
    """One of two ranks sharing the GPU over gloo: train(captured_step=True) on a DDP-wrapped model through the captured path, rank 1
    gets one batch with more boxes than the captured capacity (-> every rank must step that batch eagerly)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, ".."), os.path.join(here, "..", "custom-yolo-implmentation_amd")):
        sys.path.insert(0, os.path.abspath(p))
    import tempfile
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training import train_model as tm
    from src.training.utils_train import get_optimizer, prepare_ddp_model
    torch.manual_seed(100 + rank)                       # different initial weights: the wrapper's broadcast equalises them
    model = prepare_ddp_model(model=Model(**NANO, num_classes=80), device_id=0, config={"precision": "bfloat16"},
                              world_size=world, device="cuda")
    assert type(model).__name__ == "DistributedDataParallel"

    class Loader(list):
        sampler = None

    def batches(n, seed, big_at=None):
        g = torch.Generator().manual_seed(seed)
        out_ = Loader()
        for i in range(n):
            cnt = [300, 300] if i == big_at else [3 + (i % 3), 1 + rank]
            img = torch.randn(2, 3, 160, 160, generator=g)
            tg = [{"boxes": torch.cat([torch.rand(c, 2, generator=g) * 160, torch.rand(c, 2, generator=g) * 60 + 8,
                                       torch.randint(0, 80, (c, 1), generator=g).float()], 1)} for c in cnt]
            out_.append((img, tg))
        return out_
    tr = batches(6, 7 + rank, big_at=3 if rank == 1 else None)      # 600 boxes > 128 * 2 on rank 1 only
    va = batches(2, 50 + rank)
    opt, sched = get_optimizer(model, lr=1e-4, weight_decay=1e-4, patience=3, factor=0.5)
    seen = []
    orig = tm.CapturedTraining.step

    def spy(self, images, boxes):
        ld = orig(self, images, boxes)
        seen.append((self.captured, self.dirty))
        return ld
    tm.CapturedTraining.step = spy
    with tempfile.TemporaryDirectory() as d:
        tm.train(model=model, train_loader=tr, val_loader=va, optimizer=opt, scheduler=sched, criterion=YoloDFLQFLoss(num_classes=80),
                 initial_epoch=0, num_epochs=1, device=0, num_classes=80, rank=rank, checkpoint_dir=d, distributed_mode="ddp",
                 precision="bfloat16", conf_threshold=0.01, captured_step=True)      # opt-in with more than one rank
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()]).cpu()
    bufs = torch.cat([b.detach().float().reshape(-1) for b in model.buffers()]).cpu()
    both_w = [torch.zeros_like(flat) for _ in range(world)]
    both_b = [torch.zeros_like(bufs) for _ in range(world)]
    dist.all_gather(both_w, flat)
    dist.all_gather(both_b, bufs)
    if rank == 0:
        torch.save(dict(wdiff=float((both_w[0] - both_w[1]).abs().max()), bdiff=float((both_b[0] - both_b[1]).abs().max()),
                        finite=bool(torch.isfinite(flat).all()), seen=seen), out)
    dist.barrier()
    dist.destroy_process_group()


def _shard_rank(rank, world, port, out):
    """One of two ranks sharing the GPU over gloo: `--mode fsdp2` with `native_shard: true` end to end -- prepare_fsdp2_model,
    get_optimizer (shards the model), train() through ShardedTraining (captured graphs around the two collectives), rank 1 with
    one oversize batch (every rank steps it eagerly), validation, the gathered checkpoint."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, ".."), os.path.join(here, "..", "custom-yolo-implmentation_amd")):
        sys.path.insert(0, os.path.abspath(p))
    import tempfile
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from oracle.params import det_fill_
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training import train_model as tm
    from src.training.utils_train import get_optimizer, prepare_fsdp2_model
    m = Model(**NANO, num_classes=80)
    det_fill_(m.state_dict(), 9)                        # FSDP starts from identical weights on every rank
    model = prepare_fsdp2_model(model=m, device_id=0, config={"precision": "bfloat16", "native_shard": True}, world_size=world, device="cuda")

    class Loader(list):
        sampler = None

    def batches(n, seed, big_at=None):
        g = torch.Generator().manual_seed(seed)
        out_ = Loader()
        for i in range(n):
            cnt = [300, 300] if i == big_at else [3 + (i % 3), 1 + rank]
            img = torch.randn(2, 3, 160, 160, generator=g)
            tg = [{"boxes": torch.cat([torch.rand(c, 2, generator=g) * 160, torch.rand(c, 2, generator=g) * 60 + 8,
                                       torch.randint(0, 80, (c, 1), generator=g).float()], 1)} for c in cnt]
            out_.append((img, tg))
        return out_
    tr = batches(5, 7 + rank, big_at=3 if rank == 1 else None)
    va = batches(2, 50 + rank)
    opt, sched = get_optimizer(model, lr=1e-3, weight_decay=1e-4, patience=3, factor=0.5)
    st = model._native_shard["state"]
    before = st.master.detach().clone()
    seen = []
    orig = tm.ShardedTraining.step

    def spy(self, images, boxes):
        ld = orig(self, images, boxes)
        seen.append((self.captured, self.dirty, self.runner.graph is not None))
        return ld
    tm.ShardedTraining.step = spy
    with tempfile.TemporaryDirectory() as d:
        tm.train(model=model, train_loader=tr, val_loader=va, optimizer=opt, scheduler=sched, criterion=YoloDFLQFLoss(num_classes=80),
                 initial_epoch=0, num_epochs=1, device=0, num_classes=80, rank=rank, checkpoint_dir=d, distributed_mode="fsdp2",
                 precision="bfloat16", conf_threshold=0.01)
        dist.barrier()
        ck = torch.load(os.path.join(d, "model_epoch_1.pth"), map_location="cpu", weights_only=False) if rank == 0 else None
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()]).cpu()      # the gathered bf16 compute copies
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    moved = float((st.master.detach() - before).abs().max())
    if rank == 0:
        bare = Model(**NANO, num_classes=80)
        bare.load_state_dict(ck["model_state"])
        ck_vs_lowp = max(float((dict(bare.named_parameters())[k].float() - p.detach().float().cpu()).abs().max() /
                               p.detach().float().abs().max().clamp_min(1e-6)) for k, p in model.named_parameters() if p.requires_grad)
        torch.save(dict(wdiff=float((both[0] - both[1]).abs().max()), finite=bool(torch.isfinite(flat).all()), seen=seen, moved=moved,
                        shard_frac=st.master.numel() / st.total, ck_vs_lowp=ck_vs_lowp), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_native_sharded_training_two_ranks_sharing_the_gpu_over_gloo(tmp_path):
    """config 4's native route on N = 2 as far as a one-GPU box can go (two processes on the card, gloo instead of RCCL): both
    ranks end the epoch with identical parameters, each owns half of the master vector, the step was captured and the
    oversize batch of rank 1 sent BOTH ranks through the eager step, the gathered fp32 checkpoint matches the bf16 copies."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "shard.pt")
    mp.spawn(_shard_rank, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["finite"] and r["wdiff"] == 0.0 and r["moved"] > 0, r
    assert abs(r["shard_frac"] - 0.5) < 1e-9
    steps = r["seen"]
    assert len(steps) == 5 and steps[0][0] and steps[0][2] and steps[3][1] and not steps[4][1], steps
    assert r["ck_vs_lowp"] < 2.0 ** -7, r           # fp32 masters vs their bf16 images


@pytest.mark.timeout(600)
def test_captured_ddp_two_ranks_sharing_the_gpu_over_gloo(tmp_path):
    """The drop-in default on N = 2 (as far as a one-GPU box can go: two processes on the card, gloo instead of RCCL):
    weights and BatchNorm buffers identical across ranks after an epoch in which rank 1 alone had a batch that does not
    fit the captured buffers (all ranks must fall back to the eager step for it, with the same buckets / reduction)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "ddp.pt")
    mp.get_context("spawn")
    mp.spawn(_ddp_rank, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["finite"] and r["wdiff"] == 0.0 and r["bdiff"] == 0.0, r
    steps = r["seen"]
    assert len(steps) == 6 and steps[0][0] and steps[3][1] and not steps[4][1], steps     # captured at 0, eager at 3, replay again at 4
