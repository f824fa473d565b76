"""cProfile of bench.measure_fsdp2 (preset l under prepare_fsdp2_model, one-rank RCCL group, the reference's loop body): where
the host time of config 4's wrapper route goes.  Usage: host_profile_fsdp2.py [preset] [batch]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
sys.path.insert(0, ROOT)
import torch
import bench

preset = sys.argv[1] if len(sys.argv) > 1 else "l"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
print(bench.measure_fsdp2(preset, batch, 640, 80, dev, steps=10, warmup=3), flush=True)
pr = cProfile.Profile()
pr.enable()
print(bench.measure_fsdp2(preset, batch, 640, 80, dev, steps=5, warmup=1), flush=True)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
