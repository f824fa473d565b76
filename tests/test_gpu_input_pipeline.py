"""-m gpu: the on-device input pipeline (csrc/image_prep.hip through src/data/transforms.py::BatchTransform) against the
CPU oracle (oracle/image_prep.py) on decoded images of assorted sizes, with fixed and with sampled random decisions.

Both sides quantise to uint8 levels after the resize and after every colour op, as the reference's uint8 pipeline does;
the device accumulates the resize taps in a different order than the oracle, so a pixel that lands within rounding
distance of a level boundary may come out one level apart (and the hue round trip can turn that into two): the normalised
outputs must agree exactly on >= 99.5 % of the elements after the resize alone and >= 97 % after the four colour ops
(resampling with dyadic weights produces exact .5 ties whose rounding depends on the summation order, and every
further quantisation step can move such a pixel once more; the hue round trip scales a one-level difference of one
channel by up to the saturation gain), within 3 levels (3 / 255 / 0.224 = 0.053) after the resize and 6 after the chain."""
import time

import numpy as np
import pytest
import torch

from oracle import image_prep as oip

pytestmark = pytest.mark.gpu
SIZES = [(480, 640), (333, 500), (640, 427), (1000, 750), (64, 64), (17, 301), (640, 640), (200, 1200)]


def _images(seed, sizes):
    rng = np.random.default_rng(seed)
    out = []
    for h, w in sizes:
        # smooth content + noise: like a photograph, most pixels are not on a rounding boundary after resampling
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / (7.0 + c) + yy / 13.0) for c in range(3)], -1)
        out.append(torch.from_numpy(np.clip(base + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)))
    return out


def _compare(got, want, what, exact_min=0.995, levels=3):
    got, want = got.float().cpu(), want.float()
    d = (got - want).abs()
    exact = float((d < 1e-5).float().mean())
    lim = (levels + 0.05) / 255 / 0.224
    assert exact >= exact_min and float(d.max()) <= lim, f"{what}: {exact:.4%} exact, max {float(d.max()):.4f} (limit {lim:.4f})"


@pytest.mark.parametrize("size", [640, 320])
def test_validation_transform_resize_normalise(size):
    from src.data.transforms import BatchTransform
    imgs = _images(1, SIZES)
    batch, _ = BatchTransform(False, size=size, device="cuda")(imgs)
    assert batch.shape == (len(imgs), 3, size, size) and batch.dtype == torch.float32
    for i, im in enumerate(imgs):
        _compare(batch[i], oip.transform_image(im.numpy(), size), f"val image {i} {tuple(im.shape)}")


def test_training_transform_every_op_order_and_flip():
    from src.data.transforms import BatchTransform
    imgs = _images(2, SIZES)
    orders = [(0, 1, 2, 3), (3, 2, 1, 0), (1, 3, 0, 2), (2, 0, 3, 1), (0, 2, 1, 3), (3, 0, 1, 2), (1, 0, 2, 3), (2, 3, 0, 1)]
    params = [(i % 2 == 0, orders[i], (0.8 + 0.05 * i, 1.2 - 0.04 * i, 0.85 + 0.04 * i, -0.1 + 0.028 * i)) for i in range(len(imgs))]
    tr = BatchTransform(True, size=640, device="cuda")
    tg = [{"boxes": torch.tensor([[10., 5., 20., 10.]]), "labels": torch.tensor([[float(i)]])} for i in range(len(imgs))]
    batch, out = tr(imgs, tg, params=params)
    for i, (im, (flip, order, fac)) in enumerate(zip(imgs, params)):
        _compare(batch[i], oip.transform_image(im.numpy(), 640, flip, order, fac), f"train image {i} {tuple(im.shape)} {order} flip={flip}", 0.97, 6)
        ref = oip.transform_boxes(tg[i]["boxes"], im.shape[1], im.shape[0], 640, flip)
        assert torch.allclose(out[i]["boxes"][:, :4], ref) and float(out[i]["boxes"][0, 4]) == float(i)


def test_sampled_decisions_bf16_output_and_throughput():
    from src.data.transforms import BatchTransform
    imgs = _images(3, [(480, 640)] * 32)
    tr = BatchTransform(True, size=640, device="cuda", dtype=torch.bfloat16)
    torch.manual_seed(11)
    params = [tr.sample() for _ in imgs]
    batch, _ = tr(imgs, params=params)
    assert batch.dtype == torch.bfloat16 and torch.isfinite(batch.float()).all()
    for i in (0, 13, 31):
        want = oip.transform_image(imgs[i].numpy(), 640, *params[i]).to(torch.bfloat16)
        _compare(batch[i], want.float(), f"bf16 image {i}", 0.97, 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr(imgs, params=params)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"\n[input pipeline] 32 images 480x640 -> 640x640 bf16 incl. pinned upload: {ms:.2f} ms / batch = {32e3 / ms:.0f} img/s")


def test_device_prefetcher_yields_the_loader_s_batches_in_order_while_the_consumer_stream_is_busy():
    """DevicePrefetcher (src/data/data_loader.py): batch i+1 is uploaded on a side stream while the consumer's stream still
    works on batch i; the consumer sees every batch complete, in order, equal to the loader's own tensors -- also when it keeps
    its stream busy, drops references early and the allocator recycles upload buffers."""
    import torch
    from src.data.data_loader import DevicePrefetcher
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(8, 3, 160, 160, generator=g).pin_memory(), [{"boxes": torch.full((2, 5), float(i))} for _ in range(8)])
               for i in range(7)]
    busy_a, busy_b = torch.randn(16 << 20, device="cuda"), torch.empty(16 << 20, device="cuda")
    seen = 0
    pf = DevicePrefetcher(batches, "cuda")
    assert len(pf) == 7
    for i, (img, tg) in enumerate(pf):
        assert img.is_cuda and float(tg[0]["boxes"][0, 0]) == float(i)
        for _ in range(3):
            busy_b.copy_(busy_a)                       # the "step": the consumer stream has work queued when the next upload starts
        s = (img.double() - batches[i][0].cuda().double()).abs().max()
        assert float(s) == 0.0, (i, float(s))
        del img
        seen += 1
    assert seen == 7
    assert list(DevicePrefetcher(batches[:2], "cpu"))[1][0] is batches[1][0]      # CPU: the loader's own items, untouched
