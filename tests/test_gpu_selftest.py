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


def test_lds_dma_lane_layout_and_out_of_range_lanes():
    """buffer_load_dwordx4 ... lds: lane l's 16 bytes land at LDS base + 16*l (lane-linear), and what an out-of-range
    lane leaves in its slot is recorded here (the conv loaders rely on VGPR-destination loads returning 0; an
    LDS-DMA loader may only rely on what this test pins)."""
    import torch
    from src.hipops import lib
    src = torch.arange(128 * 4, dtype=torch.int32).view(128, 4).cuda()
    out = torch.zeros(64, 4, dtype=torch.int32, device="cuda")
    lib.call("yolo_selftest_glds", src.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu()
    for l in range(64):
        if l % 4 != 3:
            assert o[l].tolist() == src[2 * l].cpu().tolist(), (l, o[l].tolist())
    oob = o[3::4]
    kind = "zeros" if bool((oob == 0).all()) else "untouched" if bool((oob == -1431655766).all()) else "other"
    print("\\n[selftest] LDS-DMA out-of-range lanes leave:", kind, oob[0].tolist())
    assert kind in ("zeros", "untouched")


@pytest.mark.parametrize("tree", [0, 1])
def test_grid_barrier_completes_among_resident_workgroups(tree):
    """yolo_selftest_grid_barrier (the measurement kernel behind DESIGN section 6's refutation of a one-launch conv +
    BatchNorm), flat and two-level: every workgroup passes every round, nobody gives up, the arrival counters end at the
    workgroup count (top counters of the two-level form: at 16)."""
    import torch
    from src.hipops import lib
    blocks, rounds = 250, 3
    per = 17 if tree else 1
    counters = torch.zeros(rounds * per, dtype=torch.int32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib.call("yolo_selftest_grid_barrier", counters.data_ptr(), blocks, rounds, tree, flag.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    c = counters.view(rounds, per)
    if tree:
        assert c[:, 0].tolist() == [16] * rounds and c[:, 1:].sum(1).tolist() == [blocks] * rounds
    else:
        assert c[:, 0].tolist() == [blocks] * rounds


def test_permlane16_swap_row_mapping():
    """v_permlane16_swap_b32 with both operands equal: the lane ^ 16 exchange of store_pixel_blocks (conv_dev.h) without an
    LDS operation -- first result = v[lane - 16] in the odd 16-lane rows, second = v[lane + 16] in the even rows."""
    from src.hipops import lib
    v = (torch.arange(64, dtype=torch.int32) * 7 + 3).cuda()
    out = torch.zeros(128, dtype=torch.int32, device="cuda")
    lib.call("yolo_selftest_permlane16", v.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    r = out.cpu().view(64, 2)
    vc = v.cpu()
    for lane in range(64):
        odd = (lane >> 4) & 1
        assert int(r[lane, 0 if odd else 1]) == int(vc[lane ^ 16]), (lane, r[lane].tolist())
