"""N > 1 path on CPU: two gloo ranks.  The data-parallel logic (flat gradient all-reduce of TrainStepRunner,
torch DDP through prepare_ddp_model, packed scalar reduction) is device-agnostic host code; the model in
these tests is the product Model with its HIP leaves swapped for the plain-torch stand-ins."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = dict(csp=[False, True], depth=[1] * 6, width=[3, 8, 16, 16, 32, 128])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    img = torch.randn(2, 3, 64, 64, generator=g)
    gts = [torch.tensor([[20., 24., 16., 12., 1.], [40., 30., 20., 30., 2.]]), torch.tensor([[32., 32., 24., 24., 0.]])]
    return img, gts


def _worker(rank, world, port, out):
    for p in (HERE, os.path.join(HERE, ".."), os.path.join(HERE, "..", "custom-yolo-implmentation_amd")):
        sys.path.insert(0, os.path.abspath(p))
    import emulated_ops
    emulated_ops.install_plain()
    from oracle.params import det_fill_
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.distributed_setup import reduce_values
    from src.training.graph_step import TrainStepRunner
    from src.training.utils_train import prepare_ddp_model

    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    img, gts = _batch(rank)
    crit = YoloDFLQFLoss(num_classes=4)

    def fresh():
        m = Model(**TINY, num_classes=4)
        det_fill_(m.state_dict(), 1)
        return m.train()

    # (1) local gradients, no communication
    m0 = fresh()
    p, a, s = m0(img)
    crit(p, gts, a, s)[0].backward()
    local = torch.cat([q.grad.flatten() for q in m0.parameters() if q.grad is not None])
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = torch.stack(gathered).mean(0)

    # (2) TrainStepRunner: flat all-reduce, DDP semantics (mean over ranks); SGD lr 0 keeps weights to compare grads
    m1 = fresh()
    opt = torch.optim.SGD(m1.parameters(), lr=0.0)
    runner = TrainStepRunner(m1, crit, opt, precision="float32", use_graph=False)
    runner.capture(img, PackedTargets(gts, "cpu"))
    runner.step()
    g1 = torch.cat([q.grad.flatten() for q in m1.parameters() if q.grad is not None])

    # (3) torch DDP through the reference-named wrapper
    m2 = prepare_ddp_model(fresh(), 0, {"find_unused_parameters": False}, world, "cpu")
    p, a, s = m2(img)
    crit(p, gts, a, s)[0].backward()
    g2 = torch.cat([q.grad.flatten() for q in m2.parameters() if q.grad is not None])

    # (4) the same runner with ONE bucket (no stage cut): the bucketing must not change the result
    m3 = fresh()
    r3 = TrainStepRunner(m3, crit, torch.optim.SGD(m3.parameters(), lr=0.0), precision="float32", use_graph=False, buckets=1)
    r3.capture(img, PackedTargets(gts, "cpu"))
    r3.step()
    g3 = torch.cat([q.grad.flatten() for q in m3.parameters() if q.grad is not None])
    assert runner.staged and not r3.staged and len(runner.buckets.flats) == 2 and len(r3.buckets.flats) == 1

    # (5) CapturedTraining's collective decisions: the fit flag is the AND over ranks; sync_buffers leaves rank 0's
    # BatchNorm buffers everywhere (DistributedDataParallel's broadcast_buffers semantics)
    from src.training.train_model import CapturedTraining
    ct = CapturedTraining(m1, crit, opt, "float32")
    fit_and = [ct._all_fit(True), ct._all_fit(rank == 0), ct._all_fit(False)]
    with torch.no_grad():
        for b in m1.buffers():
            if b.is_floating_point():
                b.add_(float(rank + 1))
    ct.sync_buffers()
    bufs = torch.cat([b.float().flatten() for b in m1.buffers()])
    all_bufs = [torch.zeros_like(bufs) for _ in range(world)]
    dist.all_gather(all_bufs, bufs)

    red = reduce_values([float(rank), 10.0 + rank, 3.0], average=True)
    if rank == 0:
        torch.save(dict(e1=float((g1 - mean).abs().max()), e2=float((g2 - mean).abs().max()), scale=float(mean.abs().max()),
                        e3=float((g3 - mean).abs().max()), fit_and=fit_and, buf_diff=float((all_bufs[0] - all_bufs[1]).abs().max()),
                        differs=float((local - mean).abs().max()), red=red), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gradient_averaging(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["differs"] > 1e-3 * r["scale"]                 # ranks really had different gradients
    assert r["e1"] <= 1e-5 * r["scale"], r                   # runner == mean of local grads
    assert r["e2"] <= 1e-5 * r["scale"], r                   # DDP wrapper == mean of local grads
    assert r["e3"] <= 1e-5 * r["scale"], r                   # one bucket == two buckets == mean
    assert r["fit_and"] == [True, False, False]              # a batch that does not fit on ONE rank sends all ranks eager
    assert r["buf_diff"] == 0.0                              # buffers after sync_buffers: rank 0's on both ranks
    assert r["red"] == [0.5, 10.5, 3.0]


def test_single_process_wrappers_and_reduce_are_noops_without_a_group():
    sys.path.insert(0, os.path.join(HERE, "..", "custom-yolo-implmentation_amd"))
    from src.training.distributed_setup import reduce_value, reduce_values
    assert reduce_value(3.5) == 3.5 and reduce_values([1.0, 2.0]) == [1.0, 2.0]
