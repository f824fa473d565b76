"""Where the HOST time of the launch-by-launch training step goes (the reference's loop body: forward, loss, backward, optimizer
step -- what a multi-rank job runs by default): cProfile over a few eager steps.  Usage: host_profile.py [preset] [batch] [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import PRESETS, synthetic_batch
from src.model.losses import YoloDFLQFLoss
from src.model.model_builder import Model
from src.training.fused_adamw import HipAdamW

preset = sys.argv[1] if len(sys.argv) > 1 else "s"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Model(**PRESETS[preset], num_classes=80).to(dev).train()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
crit = YoloDFLQFLoss(num_classes=80)
img, gts = synthetic_batch(batch, 640, 80, 1234, dev)


def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        preds, anchors, strides = model(img)
        loss, _ = crit(preds, gts, anchors, strides)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
host = (time.perf_counter() - t0) / steps * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / steps * 1e3
print(f"preset {preset} batch {batch}: host issues a step in {host:.2f} ms, step incl. device {total:.2f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
