"""Calibration: library GEMM (torch.matmul -> hipBLASLt/rocBLAS) vs our 1x1 conv kernel on conv-shaped problems, warm clocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops

def timeit(fn, secs=0.6):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(10): fn()
        torch.cuda.synchronize(); n += 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 20

for (M, K, N) in [(204800, 1152, 128), (819200, 576, 64), (819200, 96, 128), (3276800, 32, 32), (51200, 2304, 256), (12800, 4608, 512), (204800, 512, 128),
                  (12800, 1024, 512), (12800, 768, 512), (12800, 512, 256), (12800, 256, 256), (12800, 512, 1024), (51200, 384, 256),
                  (51200, 256, 256), (51200, 768, 256)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(a, b))
    fl = 2.0 * M * K * N
    by = (M * K + M * N + K * N) * 2
    line = f"M={M:8d} K={K:5d} N={N:4d}  lib {us:8.1f} us {fl/us/1e6:7.1f} TF/s {by/us/1e3:6.0f} GB/s"
    # ours: 1x1 conv on an NHWC tensor with H*W*N = M, Cin = K
    if K % 32 == 0 and M % 32 == 0:
        n = 32; hw = M // n; h = int(hw ** 0.5); w = hw // h
        if h * w == hw:
            x = torch.randn(n, K, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
            wt = torch.randn(N, K, 1, 1, device="cuda") * 0.05
            wp = ops.pack_weights(wt, 1, 1, 0, torch.bfloat16)
            us2 = timeit(lambda: ops.conv_fwd(x, wp, None, N, 1, 1))
            line += f" | ours(1x1) {us2:8.1f} us {fl/us2/1e6:7.1f} TF/s"
    print(line, flush=True)
