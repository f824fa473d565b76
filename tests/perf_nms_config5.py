"""Inference tail on BASELINE config 5 shapes (preset l @1280: M = 33600 anchors, 80 classes, fp16): head decode +
class-aware NMS on the device vs the oracle's restatement of the reference's Python / torchvision path on the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file lives in tests/)
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, ROOT)
import torch
from src.hipops import ops
from src.utils.model_utils import non_max_suppression
from oracle import postproc as opost

torch.manual_seed(0)
BS, NC, M = 8, 80, 33600
dev = "cuda"
# boxes = random cxcywh in [0,1280]; logits arranged so that several thousand candidates pass conf_thres
pred = torch.empty(BS, 4 + NC, M)
pred[:, 0:2] = torch.rand(BS, 2, M) * 1280
pred[:, 2:4] = torch.rand(BS, 2, M) * 200 + 20
pred[:, 4:] = torch.rand(BS, NC, M) * 0.2
hot = torch.rand(BS, M) < 0.15
cls = torch.randint(0, NC, (BS, M))
pred[:, 4:].scatter_(1, cls.unsqueeze(1), (torch.rand(BS, 1, M) * 0.7 + 0.3) * hot.unsqueeze(1) + 0.1 * (~hot).unsqueeze(1))
pred16 = pred.half().to(dev)

def run():
    return non_max_suppression(pred16, conf_thres=0.25, iou_thres=0.45, nc=NC)

out = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
cand = int((pred16[:, 4:].amax(1) > 0.25).sum()) // BS
print(f"device NMS: {BS} images x {M} anchors, ~{cand} candidates / image, {sum(o.shape[0] for o in out) // BS} kept / image: "
      f"{ms:.2f} ms / batch = {ms / BS:.2f} ms / image", flush=True)
t0 = time.perf_counter()
ref = opost.non_max_suppression(pred.half().float()[:1].clone(), conf_thres=0.25, iou_thres=0.45, nc=NC)
cpu_ms = (time.perf_counter() - t0) * 1e3
print(f"host (oracle restatement of the reference path, {torch.get_num_threads()} threads): {cpu_ms:.1f} ms / image")
same = out[0].shape == ref[0].shape and bool(torch.equal(out[0].float().cpu()[:, 5], ref[0][:, 5]))
print("first image: kept classes identical to the oracle's:", same)
