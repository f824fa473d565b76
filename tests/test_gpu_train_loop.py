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
                                                     ("ddp", "bfloat16", True), ("ddp", "float32", True)])
def test_train_one_epoch_like_the_script(pg, mode, precision, captured, tmp_path):
    from src.data.data_loader import get_data_loaders
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.train_model import train
    from src.training.utils_train import get_optimizer, prepare_ddp_model, prepare_fsdp2_model
    rank, world, gpu = pg
    torch.manual_seed(0)
    model = Model(**NANO, num_classes=80)
    wrap = prepare_ddp_model if mode == "ddp" else prepare_fsdp2_model
    if captured:        # the captured loop is for an unwrapped model (single process): see CapturedTraining
        model = model.cuda()
    else:
        model = wrap(model=model, device_id=gpu, config={"precision": precision, "find_unused_parameters": False},
                     world_size=world, device="cuda")
    tr, va = get_data_loaders("/nonexistent/train", "/nonexistent/val", "", "", batch_size=4, is_test=True, device="cuda",
                              num_classes=80, res=160)
    opt, sched = get_optimizer(model, lr=1e-4, weight_decay=1e-4, patience=3, factor=0.5)
    before = [p.detach().float().clone() for p in model.parameters()][:3]
    train(model=model, train_loader=tr, val_loader=va, optimizer=opt, scheduler=sched,
          criterion=YoloDFLQFLoss(num_classes=80), initial_epoch=0, num_epochs=1, device=gpu, num_classes=80, rank=rank,
          checkpoint_dir=str(tmp_path), distributed_mode=mode, precision=precision, conf_threshold=0.01,
          captured_step=captured)
    after = [p.detach().float() for p in model.parameters()][:3]
    assert any(not torch.equal(a, b) for a, b in zip(before, after)), "parameters did not move"
    assert all(torch.isfinite(a).all() for a in after)
    ck = torch.load(os.path.join(str(tmp_path), "model_epoch_1.pth"), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and "model_state" in ck and "optimizer_state" in ck
    keys = set(ck["model_state"].keys())
    assert any(k.endswith("net.p1.0.conv.weight") for k in keys) and any(k.endswith("head.dfl.conv.weight") for k in keys)


def test_two_graph_data_parallel_step_equals_single_graph_step(pg):
    """The N > 1 step of TrainStepRunner (graph 1: fwd+bwd+pack gradients, RCCL all-reduce, graph 2: unpack + AdamW)
    on a one-rank RCCL group must move the weights exactly like the single-graph step; with bf16-compressed
    gradients it must stay within bf16 rounding of it."""
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.graph_step import TrainStepRunner
    g = torch.Generator().manual_seed(11)
    img = torch.randn(2, 3, 160, 160, generator=g).cuda()
    gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                      torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]

    def run(force, comm_dtype):
        torch.manual_seed(0)
        model = Model(**NANO, num_classes=80).cuda().train()
        # lr 0: the weights stay put, so the gradients of the last step are comparable across runs (an Adam update is
        # sign-like for near-zero gradients and would amplify the last-bit noise of the atomically summed statistics)
        opt = torch.optim.AdamW(model.parameters(), lr=0.0, weight_decay=0.0, capturable=True, fused=True)
        # fp32 compute: the only run-to-run noise left is the last bits of the atomically summed batch statistics
        r = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, "float32", use_graph=True,
                            grad_comm_dtype=comm_dtype, force_comm=force)
        r.capture(img, PackedTargets(gts, img.device), warmup=1)
        for _ in range(2):
            r.step()
        torch.cuda.synchronize()
        assert (r.graph2 is not None) == force
        st = opt.state[next(iter(model.parameters()))]
        assert int(st["step"]) >= 2                                     # the optimizer really stepped (in graph 2)
        return [p.grad.detach().float().clone() for p in model.parameters() if p.grad is not None]

    base, same, comp = run(False, None), run(True, None), run(True, torch.bfloat16)
    gmax = max(float(a.abs().max()) for a in base)
    for a, b, c in zip(base, same, comp):
        scale = a.abs().max().clamp_min(1e-3 * gmax)       # some gradients are mathematically zero (noise only)
        assert torch.isfinite(b).all() and torch.isfinite(c).all()
        assert (a - b).abs().max() / scale < 1e-3, "pack / all-reduce / unpack changed the gradients"
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

    def grads():
        return [p.grad.detach().float().clone() for p in model.parameters() if p.grad is not None]

    amp = torch.bfloat16 if precision == "bfloat16" else None
    # bf16: the same MFMA kernels in the same order on both sides; what differs is the arrival order of the float
    # atomics behind the batch statistics, i.e. last-bit noise of fp32 sums that a bf16 rounding turns into one ulp
    # (2^-8) of some activations, which later layers amplify: single elements of small gradient tensors move by a few
    # per cent of the tensor's maximum, so bf16 is held to a per-tensor relative L2 bound instead of the element-wise one
    tol = 1e-3 if amp is None else 3e-2

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
        nmax = max(float(a.norm()) for a in want)
        for a, b in zip(want, got):
            if amp is None:
                scale = a.abs().max().clamp_min(1e-3 * gmax)
                assert (a - b).abs().max() / scale < tol, (what, float((a - b).abs().max() / scale))
            else:
                rel = float((a - b).norm() / a.norm().clamp_min(1e-3 * nmax))
                assert rel < tol, (what, rel)

    want = plain(img)
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
