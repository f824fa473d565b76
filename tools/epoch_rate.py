"""train()'s own epoch loop (src/training/train_model.py::_run_epoch with the captured step) over batches that live in pinned HOST
memory, 60 steps of preset s at 32 images: images/s of the loop a user runs, upload included."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import PRESETS, synthetic_batch
from src.model.losses import YoloDFLQFLoss
from src.model.model_builder import Model
from src.training import train_model as tm
from src.training.fused_adamw import HipAdamW

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Model(**PRESETS["s"], num_classes=80).to(dev).train()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
crit = YoloDFLQFLoss(num_classes=80)
img, gts = synthetic_batch(32, 640, 80, 1234, dev)
host = [img.cpu().pin_memory() for _ in range(4)]
targets = [{"boxes": g.cpu()} for g in gts]


class Loader(list):
    sampler = None


cap = tm.CapturedTraining(model, crit, opt, "bfloat16")
kw = dict(device_type="cuda", dtype=torch.bfloat16, enabled=True)
warm = Loader((host[i % 4], targets) for i in range(6))
tm._run_epoch(model, warm, crit, "cuda", kw, 0, "warm-up", opt, captured=cap)
torch.cuda.synchronize()
steps = 60
loader = Loader((host[i % 4], targets) for i in range(steps))
t0 = time.perf_counter()
out = tm._run_epoch(model, loader, crit, "cuda", kw, 0, "epoch", opt, captured=cap)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"_run_epoch, captured step, pinned host batches: {dt / steps * 1e3:.2f} ms/step = {32 * steps / dt:.0f} img/s; mean losses {[round(v, 4) for v in out]}")
