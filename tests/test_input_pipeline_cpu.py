"""(f)-2 input pipeline on the CPU: the oracle's antialiased resize pinned against torch's own op (what torchvision's
Resize calls; torchvision itself is absent: parity unpinned for its colour arithmetic), the host logic of BatchTransform
(random draws in torchvision's order, box geometry, batch format) with the device leaf swapped for the oracle, and the
parquet dataset + loader producing collate.py's (images, [targets]) format."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import emulated_ops
from oracle import image_prep as oip


@pytest.fixture(autouse=True)
def _emulate(monkeypatch):
    emulated_ops.install(monkeypatch)


@pytest.mark.parametrize("h,w,s", [(480, 640, 640), (333, 500, 640), (1000, 750, 640), (100, 120, 64), (64, 64, 64), (17, 301, 96)])
def test_oracle_resize_matches_torch_antialiased_bilinear(h, w, s):
    img = np.random.default_rng(h * w).integers(0, 256, (h, w, 3)).astype(np.float32)
    got = oip.resize_aa(img, s)
    want = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(s, s), mode="bilinear", antialias=True,
                         align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - want).max() < 1e-3          # uint8 levels


def test_oracle_colour_ops_known_answers():
    x = np.random.default_rng(1).integers(0, 256, (9, 11, 3)).astype(np.uint8)
    ident = oip.transform_image(x, 9 if False else 11, order=(), mean=(0, 0, 0), std=(1, 1, 1))     # resize only
    assert ident.shape == (3, 11, 11)
    sq = np.random.default_rng(2).integers(0, 256, (8, 8, 3)).astype(np.uint8)
    base = oip.transform_image(sq, 8, order=(), mean=(0, 0, 0), std=(1, 1, 1)) * 255
    assert torch.equal(base.round(), torch.from_numpy(sq.astype(np.float32)).permute(2, 0, 1))     # same size: identity
    # factor 1 / hue 0 leave the image alone up to the float->uint8 rule of the hue round trip (<= 1 level)
    same = oip.transform_image(sq, 8, order=(0, 1, 2), factors=(1.0, 1.0, 1.0, 0.0), mean=(0, 0, 0), std=(1, 1, 1)) * 255
    assert torch.equal(same.round(), base.round())
    hue0 = oip.transform_image(sq, 8, order=(3,), factors=(1.0, 1.0, 1.0, 0.0), mean=(0, 0, 0), std=(1, 1, 1)) * 255
    assert float((hue0 - base).abs().max()) <= 1.0 + 1e-3
    dark = oip.transform_image(sq, 8, order=(0,), factors=(0.5, 1, 1, 0), mean=(0, 0, 0), std=(1, 1, 1)) * 255
    assert torch.equal(dark.round(), torch.floor(base.round() * 0.5))
    gray = oip.transform_image(sq, 8, order=(2,), factors=(1, 1, 0.0, 0), mean=(0, 0, 0), std=(1, 1, 1)) * 255
    assert float((gray[0] - gray[1]).abs().max()) == 0 and float((gray[1] - gray[2]).abs().max()) == 0   # saturation 0 = grayscale


def test_batch_transform_host_logic_and_box_geometry():
    from src.data.transforms import BatchTransform
    rng = np.random.default_rng(3)
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)) for h, w in ((40, 60), (33, 20), (64, 64))]
    tg = [{"boxes": torch.tensor([[10., 5., 20., 10.]]), "labels": torch.tensor([[3.]]), "name": "a"},
          {"boxes": torch.zeros(0, 4), "labels": torch.zeros(0, 1), "name": "b"},
          {"boxes": torch.tensor([[0., 0., 64., 64.], [16., 8., 8., 4.]]), "labels": torch.tensor([[1.], [2.]]), "name": "c"}]
    tr = BatchTransform(True, size=32, device="cpu")
    torch.manual_seed(5)
    params = [tr.sample() for _ in imgs]
    torch.manual_seed(5)                                     # the draws in torchvision's order, reproducible from the seed
    want = []
    for _ in imgs:
        flip = bool(torch.rand(1) < 0.5)
        order = tuple(int(i) for i in torch.randperm(4))
        want.append((flip, order, tuple(float(torch.empty(1).uniform_(a, b)) for a, b in ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1)))))
    assert params == want
    batch, out = tr(imgs, tg, params=params)
    assert batch.shape == (3, 3, 32, 32) and batch.dtype == torch.float32
    for i, (im, (flip, order, fac)) in enumerate(zip(imgs, params)):
        assert torch.equal(batch[i], oip.transform_image(im.numpy(), 32, flip, order, fac))
        h, w = im.shape[:2]
        ref = oip.transform_boxes(tg[i]["boxes"], w, h, 32, flip)
        assert torch.allclose(out[i]["boxes"][:, :4], ref) and torch.equal(out[i]["boxes"][:, 4:], tg[i]["labels"])
        assert out[i]["name"] == tg[i]["name"] and out[i]["boxes"].shape[1] == 5
    flipped = [i for i, p in enumerate(params) if p[0]]
    if 0 in flipped:                                         # x' = W - x - w, then * 32 / 60
        assert abs(float(out[0]["boxes"][0, 0]) - (60 - 30) * 32 / 60) < 1e-5
    val, _ = BatchTransform(False, size=32, device="cpu")(imgs, tg)
    assert torch.equal(val[2], oip.transform_image(imgs[2].numpy(), 32))


def test_parquet_dataset_and_loader_yield_the_collate_format(tmp_path):
    import pandas as pd
    from PIL import Image
    from src.data.data_loader import get_data_loaders
    rng = np.random.default_rng(4)
    rows = []
    os.makedirs(tmp_path / "img")
    for i, (h, w) in enumerate(((48, 64), (50, 40), (32, 32), (70, 90))):
        Image.fromarray(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).save(tmp_path / "img" / f"{i}.png")
        rows.append({"file_name": f"{i}.png", "bbox": [[1.0, 2.0, 10.0 + i, 12.0]] * (i + 1), "category_id": [float(i)] * (i + 1), "name": f"im{i}"})
    pd.DataFrame(rows).to_parquet(tmp_path / "train.parquet")
    pd.DataFrame(rows[:2]).to_parquet(tmp_path / "val.parquet")
    tr, va = get_data_loaders(str(tmp_path / "train.parquet"), str(tmp_path / "val.parquet"), str(tmp_path / "img"), str(tmp_path / "img"),
                              batch_size=2, is_test=True, device="cpu", res=32)
    assert len(tr) == 2 and len(va) == 1
    for images, targets in tr:
        assert images.shape == (2, 3, 32, 32) and len(targets) == 2
        for t in targets:
            assert t["boxes"].shape[1] == 5 and t["boxes"].dtype == torch.float32 and "labels" not in t
    images, targets = next(iter(va))
    # the dataset shuffles its rows like the reference's (`df.sample(frac=...)`, global numpy RNG): either order is right
    assert torch.isfinite(images).all() and sorted(t["name"] for t in targets) == ["im0", "im1"]
