"""Hardware probe: LDS-DMA in half of the workgroups, a cross-lane exchange instruction in the other half (co-resident on the
same CUs) -- does the exchange corrupt the other workgroups' LDS-DMA data?  Prints mismatching dwords per spam kind."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import lib

pattern = torch.randint(0, 2 ** 31 - 1, (1024,), dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
names = {0: "VALU only", 1: "ds_bpermute_b32", 2: "v_permlane16_swap_b32", 3: "ds_write + ds_read"}
for rep in range(2):
    for spam in (1, 0, 2, 3, 1):
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        lib.call("yolo_selftest_dma_vs_xlane", pattern.data_ptr(), 4096, 200, spam, err.data_ptr(), sink.data_ptr(),
                 torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print(f"other workgroups run {names[spam]:24s}: {int(err.item())} mismatching dwords in 2048 x 200 LDS-DMA tiles of 4 KB", flush=True)

print("-- the three-slot ring (counted vmcnt + raw s_barrier, reads of other waves' pieces) beside the same traffic")
steps = 2000
src = torch.randint(1, 2 ** 31 - 1, (steps * 1024,), dtype=torch.int32, device="cuda")
for rep in range(2):
    for spam in (1, 0, 2, 3, 1):
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        lib.call("yolo_selftest_ring_vs_xlane", src.data_ptr(), steps, 2048, spam, err.data_ptr(), sink.data_ptr(),
                 torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print(f"other workgroups run {names[spam]:24s}: {int(err.item())} mismatching dwords in 1024 rings x {steps} steps", flush=True)
