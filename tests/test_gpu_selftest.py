"""-m gpu: instruction-semantics self-tests (what the transposed-LDS-read kernels assume about gfx950)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ds_read_b64_tr_b16_lane_mapping():
    from src.hipops import lib
    tile = torch.arange(32 * 16, dtype=torch.int16).view(32, 16)            # T[k][c] = 16k + c
    out = torch.full((64, 8), -1, dtype=torch.int16, device="cuda")
    lib.call("yolo_selftest_tr16", tile.cuda().data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu()
    want = torch.empty(64, 8, dtype=torch.int16)
    for l in range(64):
        g, i = l >> 4, l & 15
        for q in range(8):
            want[l, q] = tile[8 * g + q, i]
    print("\nlane 0:", got[0].tolist(), "lane 1:", got[1].tolist(), "lane 17:", got[17].tolist(), "lane 63:", got[63].tolist())
    assert torch.equal(got, want), "tr16 lane mapping differs from the documented one"
