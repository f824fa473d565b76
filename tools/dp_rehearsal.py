"""Rehearsal of the N > 1 TrainStepRunner path: 2 gloo ranks sharing cuda:0; prints loss / finiteness per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch, torch.distributed as dist
from src.model.losses import PackedTargets, YoloDFLQFLoss
from src.model.model_builder import Model
from src.training.graph_step import TrainStepRunner

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="env://", world_size=world, rank=rank)
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
torch.manual_seed(0)
model = Model(**NANO, num_classes=80).cuda().train()
g = torch.Generator().manual_seed(5 + rank)
img = torch.randn(2, 3, 160, 160, generator=g).cuda()
gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                  torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True, fused=True)
comm = None if os.environ.get("COMM") == "fp32" else torch.bfloat16
r = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, "bfloat16", use_graph=os.environ.get("GRAPH", "1") == "1",
                    grad_comm_dtype=comm, force_comm=True)
r.capture(img, PackedTargets(gts, img.device), warmup=2)
fin = lambda ts: all(bool(torch.isfinite(t).all()) for t in ts)
print(rank, "after capture: params finite", fin(model.parameters()), flush=True)
mode = os.environ.get("MODE", "step")
for i in range(4):
    if mode == "step":
        loss = r.step()
    else:
        r.graph.replay()
        if os.environ.get("SYNC"): torch.cuda.synchronize()
        if not os.environ.get("NOREDUCE"): r._reduce_flat()
        if os.environ.get("SYNC"): torch.cuda.synchronize()
        if mode == "eager2":
            r._unpack_grads(); opt.step()
        elif mode == "g2":
            r.graph2.replay()
        loss = r.loss
    torch.cuda.synchronize()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    print(rank, i, "loss", float(loss), "grads finite", fin(grads), "params finite", fin(model.parameters()),
          "flat finite", None if r.flat is None else bool(torch.isfinite(r.flat.float()).all()), flush=True)
dist.destroy_process_group()
