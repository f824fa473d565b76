"""CPU baseline of the hot path (test infrastructure; timed by bench.py's cpu_baseline leg only).

One training step the way the reference's CPU path runs it (slurm/distributed_training_cpu.sbatch:87-91 ->
scripts/distributed_training.py --device cpu --mode ddp --precision float32, world_size 1): fp32 forward,
DFL/QFL loss, backward, AdamW(lr 1e-4, wd 1e-4) -- on the oracle restatement, all host threads
(src/utils/common.py:25-43 keeps torch's default for world_size 1)."""
import time

import torch

from . import blocks as ob
from . import loss as ol
from .params import ParamStore


def synthetic_batch(n, res, nc=80, seed=1234, device="cpu"):
    """randn images; 1..20 boxes per image, centres U(0,res), sizes U(8, 0.4 res + 8), class U{0..nc-1}."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n, 3, res, res, generator=g)
    gts = []
    for _ in range(n):
        m = int(torch.randint(1, 21, (1,), generator=g))
        gts.append(torch.cat([torch.rand(m, 2, generator=g) * res, torch.rand(m, 2, generator=g) * (0.4 * res) + 8,
                              torch.randint(0, nc, (m, 1), generator=g).float()], 1))
    return img.to(device), [t.to(device) for t in gts]


def _gloo_world1():
    """The reference's CPU launch is `torchrun --nproc_per_node=1 ... --device cpu --mode ddp`
    (slurm/distributed_training_cpu.sbatch:87-91): a gloo process group of one rank (distributed_setup.py:19) whose DDP
    reducer all-reduces the gradients every step.  Returns (group, owned)."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        return (dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")), False
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # gloo's transport prints its connection banner to STDOUT; bench.py's stdout carries exactly one JSON line
    import os
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return dist.group.WORLD, True


def time_cpu_steps(preset="s", res=640, batch=2, steps=6, warmup=1, nc=80, budget_s=25.0, ddp_gloo=True, threads=None):
    """fp32 training steps of the oracle restatement launched like the reference's CPU job: one rank, gloo, DDP
    semantics (the gradients go through one flat all-reduce on the one-rank group), torch's default thread count
    (src/utils/common.py:25-43: world_size 1 keeps all host threads)."""
    import torch.distributed as dist
    cfg = ob.PRESETS[preset]
    default_threads = torch.get_num_threads()
    if threads is not None:
        torch.set_num_threads(threads)
    group, owned = _gloo_world1() if ddp_gloo else (None, False)
    ps = ParamStore(0, requires_grad=True)
    img, gts = synthetic_batch(batch, res, nc)
    ob.model_forward(ps, img[:1, :, :64, :64], cfg["width"], cfg["depth"], cfg["csp"], nc)      # materialise params
    params = [t for t in ps.values() if t.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4)

    def step():
        opt.zero_grad()
        preds, a, s = ob.model_forward(ps, img, cfg["width"], cfg["depth"], cfg["csp"], nc, training=True)
        tot, _, _ = ol.dfl_qfl_loss(preds, gts, a, s, nc)
        tot.backward()
        if group is not None:                       # DistributedDataParallel's gradient averaging, world size 1
            flat = torch.cat([p.grad.reshape(-1) for p in params if p.grad is not None])
            dist.all_reduce(flat, group=group)
            off = 0
            for p in params:
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
                    off += p.numel()
        opt.step()
        return float(tot)

    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    done = 0
    while done < steps and (time.perf_counter() - t0) < budget_s:
        step()
        done += 1
    dt = time.perf_counter() - t0
    if owned:
        dist.destroy_process_group()
    used = torch.get_num_threads()
    torch.set_num_threads(default_threads)
    return dict(value=batch * done / max(dt, 1e-9), unit="images/s", cores=used, kind="port",
                sample=f"{done} fp32 train steps (fwd+loss+bwd+grad all-reduce on a 1-rank gloo group+AdamW) of preset "
                       f"'{preset}' @{res}x{res}, batch {batch}, oracle restatement on {used} host threads, {dt:.1f} s")


def config5_nms_tensor(bs=8, nc=80, m=33600, seed=0):
    """BASELINE config 5's NMS / IoU stress tensor: (bs, 4+nc, 33600), boxes random cxcywh on a 1280 canvas,
    ~15 % of the anchors confident (several thousand candidates per image after conf_thres 0.25, >= 300 kept)."""
    g = torch.Generator().manual_seed(seed)
    pred = torch.empty(bs, 4 + nc, m)
    pred[:, 0:2] = torch.rand(bs, 2, m, generator=g) * 1280
    pred[:, 2:4] = torch.rand(bs, 2, m, generator=g) * 200 + 20
    pred[:, 4:] = torch.rand(bs, nc, m, generator=g) * 0.2
    hot = torch.rand(bs, m, generator=g) < 0.15
    cls = torch.randint(0, nc, (bs, m), generator=g)
    val = (torch.rand(bs, 1, m, generator=g) * 0.7 + 0.3) * hot.unsqueeze(1) + 0.1 * (~hot).unsqueeze(1)
    pred[:, 4:].scatter_(1, cls.unsqueeze(1), val)
    return pred


def time_cpu_nms(images=2, pred=None):
    """ms / image of the reference's non_max_suppression path (Python loop + greedy NMS) restated on the host, config-5 tensor
    (`pred`: the caller's tensor, so the device leg and this one see the same input)."""
    from . import postproc as opost
    pred = (config5_nms_tensor(bs=images) if pred is None else pred[:images]).half().float()
    t0 = time.perf_counter()
    out = opost.non_max_suppression(pred.clone(), conf_thres=0.25, iou_thres=0.45, nc=80)
    dt = time.perf_counter() - t0
    return dict(value=round(1e3 * dt / images, 1), unit="ms/image", cores=torch.get_num_threads(), kind="port",
                sample=f"{images} images x 33600 anchors x 80 classes, {sum(o.shape[0] for o in out) // images} kept / image")
