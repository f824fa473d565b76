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


def time_cpu_steps(preset="s", res=640, batch=2, steps=6, warmup=1, nc=80, budget_s=25.0):
    cfg = ob.PRESETS[preset]
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
    return dict(value=batch * done / dt, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{done} fp32 train steps (fwd+loss+bwd+AdamW) of preset '{preset}' @{res}x{res}, batch {batch}, "
                       f"oracle restatement on {torch.get_num_threads()} host threads, {dt:.1f} s")
