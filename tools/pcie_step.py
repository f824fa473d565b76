"""The captured training step fed from HOST memory (a fresh pinned 32 x 3 x 640 x 640 fp32 batch per step, 157 MB): upload
queued on the step's stream against upload by DevicePrefetcher on a second stream beside the previous step, against the
device-resident batch bench.py times.  The PCIe-inclusive figures of DESIGN section 5."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import PRESETS, synthetic_batch
from src.data.data_loader import DevicePrefetcher
from src.model.losses import PackedTargets, YoloDFLQFLoss
from src.model.model_builder import Model
from src.training.fused_adamw import HipAdamW
from src.training.graph_step import TrainStepRunner

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Model(**PRESETS["s"], num_classes=80).to(dev).train()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
img, gts = synthetic_batch(32, 640, 80, 1234, dev)
runner = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, "bfloat16", use_graph=True)
runner.capture_for_batches(img, gts, warmup=1)
host = [img.cpu().pin_memory() for _ in range(4)]
cpu_gts = [g.cpu() for g in gts]
steps = 40
for _ in range(5):
    runner.step()
torch.cuda.synchronize()


def timed(fn):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def resident(n):
    for _ in range(n):
        runner.step()


def serial(n):
    for i in range(n):
        d = host[i % 4].to(dev, non_blocking=True)
        runner.step_batch(d, cpu_gts)


def prefetched(n):
    for d, _ in DevicePrefetcher([(host[i % 4], None) for i in range(n)], dev):
        runner.step_batch(d, cpu_gts)


for name, fn in (("batch resident on the device (bench.py)", resident), ("upload queued on the step's stream", serial),
                 ("upload on a second stream beside the previous step (DevicePrefetcher)", prefetched)):
    ms = timed(fn)
    print(f"{name:72s}: {ms:6.2f} ms/step = {32e3 / ms:6.0f} img/s", flush=True)
