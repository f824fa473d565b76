"""What a device-wide barrier costs inside a kernel on this GPU: a launch of B workgroups doing R barriers (arrival counter,
relaxed polling) against the same launch doing none, 20 launches per replayed graph.  The price a one-launch
"conv epilogue -> grid barrier -> normalise" kernel pays per layer, to set against the ~5 us floor of the separate BatchNorm
launch it would remove."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import lib, ops

REP = 20
dev = "cuda"
flag = torch.zeros(1, dtype=torch.int32, device=dev)


def measure(blocks, rounds, tree):
    per = (17 if tree else 1) * max(rounds, 1)
    counters = torch.zeros(REP * per, dtype=torch.int32, device=dev)

    def body():
        ops.zero_(counters)
        for i in range(REP):
            lib.call("yolo_selftest_grid_barrier", counters.data_ptr() + 4 * i * per, blocks, rounds, tree, flag.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
    body(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(7):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    return best


for tree in (0, 1):
    print("one arrival counter per barrier" if not tree else "16 group counters + a top counter per barrier (arrivals spread over 16 addresses)")
    for blocks in (64, 128, 256, 512, 1024):
        t = [measure(blocks, r, tree) for r in (0, 1, 2, 4)]
        print(f"{blocks:5d} workgroups: launch alone {t[0]:6.2f} us; + 1 barrier {t[1]:6.2f} (+{t[1] - t[0]:.2f}); + 2 {t[2]:6.2f}; "
              f"+ 4 {t[3]:6.2f} -> {(t[3] - t[0]) / 4:.2f} us per barrier; timed out: {int(flag.item())}", flush=True)
