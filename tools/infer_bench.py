"""Config 5's model half (preset l, 1280 x 1280, fp16, fused BatchNorm): forward + decode, eager launches against one replayed
hipGraph.  Usage: infer_bench.py [preset] [res] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import PRESETS
from src.hipops import ops
from src.model.model_builder import Model

preset = sys.argv[1] if len(sys.argv) > 1 else "l"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Model(**PRESETS[preset], num_classes=80).to(dev).eval().fuse()
img = torch.randn(batch, 3, res, res, device=dev)


def fwd():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        preds, anchors, strides = model(img)
        return ops.head_decode(preds, anchors, strides, 80)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


eager = timed(fwd)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd(); fwd()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = fwd()
graph = timed(g.replay)
print(f"preset {preset} {res}x{res} fp16 batch {batch}: eager {eager:.2f} ms ({eager / batch:.3f} / image), graph {graph:.2f} ms "
      f"({graph / batch:.3f} / image)", flush=True)
