"""Validation path on one 32-image batch of preset-s shapes: device select + match vs the per-image
Python loops of the reference's algorithm (oracle restatement) on the host.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file lives in tests/)
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, ROOT)
import torch
from src.hipops import ops
from src.training.metrics import DetectionMetrics
from src.training.train_model import decode_predictions_packed

torch.manual_seed(0)
N, NC, M = 32, 80, 8400
dev = "cuda"
preds = torch.randn(N, 64 + NC, M, device=dev).to(torch.bfloat16)
preds[:, 64:] -= 2.0
lv = [(80, 8), (40, 16), (20, 32)]
anchors = torch.cat([torch.stack(torch.meshgrid(torch.arange(s) + 0.5, torch.arange(s) + 0.5, indexing="ij")[::-1], 0).reshape(2, -1) for s, _ in lv], 1).to(dev)
strides = torch.cat([torch.full((1, s * s), float(st)) for s, st in lv], 1).to(dev)
gts = [torch.cat([torch.rand(k, 2) * 600 + 20, torch.rand(k, 2) * 200 + 10, torch.randint(0, NC, (k, 1)).float()], 1).to(dev)
       for k in torch.randint(1, 21, (N,)).tolist()]
m = DetectionMetrics(NC, 0.5)

def step():
    rows, count = decode_predictions_packed(preds, anchors, strides, conf_threshold=0.25, top_k=100)
    m.update_batch(rows, count, gts)

for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / 50 * 1e3
print(f"device: decode + select + match, {N} images: {gpu_ms:.3f} ms/batch ({N / gpu_ms * 1e3:.0f} img/s)", flush=True)
res = m.compute()
print({k: res[k] for k in ("true_positives", "false_positives", "false_negatives")})

# host: the reference's algorithm (oracle restatement: per-image torch ops + Python matching loops), 4 images
from oracle import postproc as opost
pc, ac, sc = preds[:4].float().cpu(), anchors.cpu(), strides.cpu()
t0 = time.perf_counter()
dec = opost.decode_predictions(pc, ac, sc, conf_threshold=0.25, top_k=100)
mc = opost.MetricCounters(NC, 0.5)
for p, g in zip(dec, gts[:4]):
    mc.update(p, g.cpu())
cpu_ms = (time.perf_counter() - t0) / 4 * 1e3
print(f"host (oracle port of the reference loops, {torch.get_num_threads()} threads): {cpu_ms:.1f} ms/image ({1e3 / cpu_ms:.1f} img/s)")
