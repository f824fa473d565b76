"""-m gpu: the validation path on the device against the reference's own outputs.

* yolo_val_select (decode_predictions' per-image selection, train_model.py:14-142) against the golden the reference
  produced (tests/golden/decode_val.npz: fp32, conf 0.6, top_k 10) and against the oracle on larger seeded inputs;
* yolo_val_match (DetectionMetrics.update, metrics.py:68-157) against the counters the reference produced
  (tests/golden/metrics.npz) -- integers, compared exactly -- and against the oracle on random batches.
Selection ties: torch.topk leaves the order of equal scores open; the kernel takes the lower anchor first, the oracle
comparison uses the same stable order (only reachable when more than top_k anchors pass and scores collide)."""
import numpy as np
import pytest
import torch

import emulated_ops as emu
from conftest import load_golden
from oracle import postproc as opost

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_decode_predictions_matches_reference_golden():
    from src.training.train_model import decode_predictions, decode_predictions_packed
    gd = load_golden("decode_val")
    args = (gd["preds"].to(DEV), gd["anchors"].to(DEV), gd["strides"].to(DEV))
    out = decode_predictions(*args, conf_threshold=0.6, top_k=10)
    for i in range(2):
        want = gd[f"out{i}"].reshape(-1, 5)
        got = out[i].cpu()
        assert got.shape == want.shape and got.dtype == torch.float32
        assert torch.equal(got[:, 4], want[:, 4]), "classes / order"
        assert torch.allclose(got[:, :4], want[:, :4], rtol=1e-5, atol=1e-4)
    rows, count = decode_predictions_packed(*args, conf_threshold=0.6, top_k=10)
    assert rows.shape == (2, 10, 6) and count.tolist() == [o.shape[0] for o in out]
    assert float(rows[0, int(count[0]):].abs().sum()) == 0.0            # zero padded
    assert bool((rows[0, :int(count[0]), 5] >= 0.6).all())              # scores ride along in column 5


@pytest.mark.parametrize("dtype,conf,top_k", [(torch.float32, 0.25, 100), (torch.bfloat16, 0.25, 100),
                                               (torch.float16, 0.5, 7), (torch.float32, 0.999, 100)])
def test_val_select_vs_oracle(dtype, conf, top_k):
    from src.hipops import ops
    g = torch.Generator().manual_seed(5)
    n, nc, m = 3, 80, 2100
    y = torch.randn(n, 4 + nc, m, generator=g)
    y[:, :4] = y[:, :4].abs() * 100
    y[:, 4:] = y[:, 4:] * 1.5 - 2.5
    y[2, 4:] = -9.0                                                     # an image with no survivor
    y = y.to(dtype)
    rows, count = ops.val_select(y.to(DEV), nc, conf, top_k)
    want_rows, want_count = emu.val_select(y, nc, conf, top_k)
    assert count.cpu().tolist() == want_count.tolist()
    assert int(count[2]) == 0
    got = rows.cpu()
    # classes and boxes exact (copied values); the sigmoid may differ from the CPU's by an ulp in fp32
    assert torch.equal(got[..., :5], want_rows[..., :5])
    assert torch.allclose(got[..., 5], want_rows[..., 5], rtol=1e-6 if dtype == torch.float32 else 0, atol=0)


def _cases(gd):
    po, go = gd["pred_off"], gd["gt_off"]
    return [(gd["pred"][po[i]:po[i + 1]], gd["gt"][go[i]:go[i + 1]]) for i in range(len(po) - 1)]


@pytest.mark.parametrize("thr", [0.5, 0.45])
def test_detection_metrics_match_reference_counters(thr):
    from src.training.metrics import DetectionMetrics
    gd = load_golden("metrics")
    nc = int(gd["num_classes"])
    tag = f"thr{thr}"
    m = DetectionMetrics(nc, iou_threshold=thr)
    for i, (p, t) in enumerate(_cases(gd)):                              # the reference's per-image API
        m.update(p.to(DEV), t.to(DEV))
        m.sync()
        got = [m.total_predictions, m.total_ground_truths, m.true_positives, m.false_positives, m.false_negatives]
        assert got == gd[tag + ":scalars_after_each"][i].tolist(), f"image {i}"
    for name, arr in (("class_tp", m.class_tp), ("class_fp", m.class_fp), ("class_fn", m.class_fn),
                      ("class_gt", m.class_gt_count)):
        assert arr.tolist() == gd[f"{tag}:{name}"].long().tolist(), name
    res = m.compute()
    for k, v in zip(gd[tag + ":compute_keys"], gd[tag + ":compute_vals"].tolist()):
        assert abs(res[str(k)] - v) <= 1e-12 + 1e-7 * abs(v), (k, res[str(k)], v)
    cm = m.get_class_metrics(2)
    for k, v in zip(gd[tag + ":class2_keys"], gd[tag + ":class2_vals"].tolist()):
        assert abs(cm[str(k)] - v) <= 1e-12 + 1e-6 * abs(v), (k, cm[str(k)], v)


def test_detection_metrics_batched_equals_per_image_and_oracle():
    """one launch for the whole batch (padded rows, concatenated targets) == the reference's counters; images
    without targets are skipped only when asked to (the validation loop's rule)"""
    from src.training.metrics import DetectionMetrics
    gd = load_golden("metrics")
    nc = int(gd["num_classes"])
    cases = _cases(gd)
    k = max(p.shape[0] for p, _ in cases)
    rows = torch.zeros(len(cases), k, 6)
    count = torch.zeros(len(cases), dtype=torch.int32)
    for i, (p, _) in enumerate(cases):
        rows[i, :p.shape[0], :5] = p
        count[i] = p.shape[0]
    for skip in (False, True):
        m = DetectionMetrics(nc, 0.5)
        m.update_batch(rows.to(DEV), count.to(DEV), [t.to(DEV) for _, t in cases], skip_empty_targets=skip)
        want = opost.MetricCounters(nc, 0.5)
        for p, t in cases:
            if skip and t.shape[0] == 0:
                continue
            want.update(p, t)
        m.sync()
        got = [m.total_predictions, m.total_ground_truths, m.true_positives, m.false_positives, m.false_negatives]
        assert got == want.scalars(), skip
        assert m.class_fp.tolist() == want.class_fp.tolist() and m.class_fn.tolist() == want.class_fn.tolist()
    assert got == gd["thr0.5:scalars_after_each"][-1].tolist() or skip      # skip=False run equals the reference


def test_val_match_random_batches_vs_oracle():
    """64 images, up to 100 predictions and up to 150 targets (more than one wave of targets), 80 classes"""
    from src.hipops import ops
    g = torch.Generator().manual_seed(11)
    n, k, nc = 64, 100, 80
    rows = torch.zeros(n, k, 6)
    count = torch.randint(0, k + 1, (n,), generator=g).to(torch.int32)
    sizes = torch.randint(0, 151, (n,), generator=g).tolist()
    sizes[3], count[5] = 0, 0
    gts = []
    for b in range(n):
        t = torch.cat([torch.rand(sizes[b], 2, generator=g) * 600, torch.rand(sizes[b], 2, generator=g) * 150 + 10,
                       torch.randint(0, 6, (sizes[b], 1), generator=g).float()], 1)
        gts.append(t)
        c = int(count[b])
        if c and sizes[b]:
            src = torch.randint(0, sizes[b], (c,), generator=g)
            p = t[src].clone()
            p[:, :2] += (torch.rand(c, 2, generator=g) - 0.5) * 0.5 * p[:, 2:4]
            rows[b, :c, :5] = p
        elif c:
            rows[b, :c, :5] = torch.rand(c, 5, generator=g) * 50
    off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32)
    ctr = torch.zeros(5 + 4 * nc, dtype=torch.int64, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.val_match(rows.to(DEV), count.to(DEV), torch.cat(gts).to(DEV), off.to(DEV), 0.5, nc, False, ctr, status)
    want = torch.zeros(5 + 4 * nc, dtype=torch.int64)
    emu.val_match(rows, count, torch.cat(gts), off, 0.5, nc, False, want, None)
    assert int(status.cpu()[0]) == 0
    assert torch.equal(ctr.cpu(), want)
    assert int(want[2]) > 500                                           # the case exercises real matches


def test_validation_epoch_runs_on_device():
    """decode_predictions_packed + update_batch on a real model output: counters are consistent"""
    from src.model.model_builder import Model
    from src.training.metrics import DetectionMetrics
    from src.training.train_model import decode_predictions_packed
    torch.manual_seed(0)
    model = Model(width=[3, 16, 32, 64, 128, 256], depth=[1] * 6, csp=[False, True], num_classes=80).to(DEV).eval()
    with torch.no_grad():
        preds, anchors, strides = model(torch.randn(2, 3, 320, 320, device=DEV))
    rows, count = decode_predictions_packed(preds, anchors, strides, conf_threshold=0.0, top_k=100)
    assert count.tolist() == [100, 100]                                 # every anchor passes: top-100 by score
    assert bool((rows[0, :-1, 5] >= rows[0, 1:, 5]).all())              # descending
    gts = [torch.tensor([[100., 100., 50., 50., 3.]], device=DEV), torch.zeros(0, 5, device=DEV)]
    m = DetectionMetrics(80, 0.5)
    m.update_batch(rows, count, gts)
    r = m.compute()
    assert r["total_ground_truths"] == 1 and r["total_predictions"] == int(count[0])
    assert r["true_positives"] + r["false_positives"] == r["total_predictions"]
    assert r["true_positives"] + r["false_negatives"] == 1
