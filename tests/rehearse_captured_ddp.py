"""Rehearsal (not collected by pytest): `train(captured_step=True)` on 2 ranks sharing one GPU over gloo -- the unwrapped
model stepped by TrainStepRunner with its own gradient all-reduce must keep the ranks' weights identical.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tests/rehearse_captured_ddp.py"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="env://", world_size=world, rank=rank)
from src.data.data_loader import get_data_loaders
from src.model.losses import YoloDFLQFLoss
from src.model.model_builder import Model
from src.training.train_model import train
from src.training.utils_train import get_optimizer, prepare_ddp_model

torch.manual_seed(100 + rank)                       # different initial weights: the broadcast must equalise them
model = Model(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256], num_classes=80)
model = prepare_ddp_model(model=model, device_id=0, config={"precision": "bfloat16", "captured_step": True}, world_size=world,
                          device="cuda")
assert type(getattr(model, "module", model)) is Model     # DDP-wrapped since round 2 (the runner steps the inner module)
tr, va = get_data_loaders("/nonexistent/train", "/nonexistent/val", "", "", batch_size=4, is_test=True, device="cuda",
                          num_classes=80, res=160)
opt, sched = get_optimizer(model, lr=1e-4, weight_decay=1e-4, patience=3, factor=0.5)
with tempfile.TemporaryDirectory() as d:
    train(model=model, train_loader=tr, val_loader=va, optimizer=opt, scheduler=sched, criterion=YoloDFLQFLoss(num_classes=80),
          initial_epoch=0, num_epochs=1, device=0, num_classes=80, rank=rank, checkpoint_dir=d, distributed_mode="ddp",
          precision="bfloat16", conf_threshold=0.01, captured_step=True)
flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()]).cpu()
both = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
diff = float((both[0] - both[1]).abs().max())
fin = bool(torch.isfinite(flat).all())
if rank == 0:
    print(f"[rehearsal] max weight difference between the two ranks after one captured epoch: {diff:.3e}; finite: {fin}")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if (diff == 0.0 and fin) else 1)
