"""Hardware probe: is `s_waitcnt vmcnt(N)` a safe COUNTED wait for LDS-DMA when the younger transfers have nothing to fetch
(zero-size descriptor / out-of-range offset -- the "issue the same number of pieces past the end of K" idiom of the ring
kernels)?  Counts lanes that read the LDS preset after vmcnt(1) although the OLDER transfer should have landed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import lib

src = torch.randint(1, 2 ** 31 - 1, ((256 << 20) // 4,), dtype=torch.int32, device="cuda")      # 256 MB: cold lines for every wave
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
names = {0: "younger = a second real load", 1: "younger = zero-size descriptor", 2: "younger = out-of-range offset"}
for rep in range(3):
    for kind in (0, 1, 2):
        flush.zero_()                                       # evict src from L2 / the memory-side cache
        err = torch.zeros(2, dtype=torch.int32, device="cuda")
        lib.call("yolo_selftest_dma_order", src.data_ptr(), src.numel() * 4, 8192, kind, err.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        e = err.tolist()
        print(f"{names[kind]:34s}: {e[0]:8d} of {8192 * 256} lanes read the preset after vmcnt(1) (older transfer not landed), {e[1]} other mismatches", flush=True)
