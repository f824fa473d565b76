"""Host-side wiring of the product (autograd Functions, nn.Modules, state-dict names) checked on CPU
by swapping the HIP leaves for plain-torch stand-ins (tests/emulated_ops.py).  The numbers are compared
with the vectors the REFERENCE produced (tests/golden), so this also pins module structure and
parameter naming.  The HIP kernels themselves are covered by the -m gpu tests."""
import pytest
import torch

import emulated_ops
from conftest import load_golden
from oracle.params import det_fill_
from test_oracle_golden import close


@pytest.fixture(autouse=True)
def _emulate(monkeypatch):
    emulated_ops.install(monkeypatch)


def _blocks():
    from torch import nn
    from src.model import model_blocks as mb
    S, I = nn.SiLU, nn.Identity
    return {
        "conv3x3s2": lambda: mb.Conv(8, 16, S(), k=3, s=2, p=1),
        "conv1x1_id": lambda: mb.Conv(16, 24, I()),
        "convdw": lambda: mb.Conv(16, 16, S(), k=3, p=1, g=16),
        "residual": lambda: mb.Residual(16),
        "c3k": lambda: mb.C3K(16, 16),
        "c3k2_res": lambda: mb.C3K2(16, 32, 1, False, 4),
        "c3k2_csp": lambda: mb.C3K2(32, 32, 2, True, 2),
        "sppf": lambda: mb.SPPF(16, 16),
        "attention": lambda: mb.Attention(128, 2),
        "psablock": lambda: mb.PSABlock(128, 2),
        "psa": lambda: mb.PSA(256, 1),
    }


@pytest.mark.parametrize("tag", ["conv3x3s2", "conv1x1_id", "convdw", "residual", "c3k", "c3k2_res", "c3k2_csp",
                                 "sppf", "attention", "psablock", "psa"])
def test_block_wiring(tag):
    gd = load_golden("block_" + tag)
    m = torch.nn.ModuleDict({"m": _blocks()[tag]()})
    det_fill_(m.state_dict(), int(gd["seed"]))
    x = gd["x"].clone().requires_grad_(True)
    m.train()
    y = m["m"](x)
    close(y, gd["y_train"], rtol=1e-4, atol=1e-5, what="y_train")
    y.backward(gd["dy"])
    close(x.grad, gd["dx"], rtol=1e-3, atol=1e-4, what="dx")
    params = dict(m.named_parameters())
    bufs = dict(m.named_buffers())
    for k, v in gd.items():
        if k.startswith("grad:"):
            close(params[k[5:]].grad, v, rtol=1e-3, atol=1e-4, what=k)
        if k.startswith("buf:"):
            close(bufs[k[4:]], v, rtol=1e-4, atol=1e-5, what=k)
    m.eval()
    with torch.no_grad():
        close(m["m"](gd["x"]), gd["y_eval"], rtol=1e-4, atol=1e-5, what="y_eval")


def test_state_dict_keys_match_reference_names():
    from src.model.model_builder import Model
    gd = load_golden("model_n320_train")
    model = Model(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256], num_classes=80)
    params = {k for k, _ in model.named_parameters()}
    want = {str(k) for k in gd["gradnorms_keys"]}          # every trainable parameter of the reference
    assert want <= params and params - want == {"head.dfl.conv.weight"}
    sd = model.state_dict()
    assert "net.p1.0.norm.num_batches_tracked" in sd and "head.cls.2.3.norm.running_var" in sd
    assert sum(p.numel() for p in model.parameters()) == 2624080


def test_model_n320_train_step_wiring():
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    gd = load_golden("model_n320_train")
    l3 = load_golden("loss_n320")
    model = Model(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256], num_classes=80)
    det_fill_(model.state_dict(), int(gd["seed"]))
    img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(50))
    model.train()
    preds, a, s = model(img)
    assert preds.shape == (2, 144, 2100) and a.shape == (2, 2100) and s.shape == (1, 2100)
    close(preds[:, :, ::7], gd["preds_stride7"], rtol=1e-3, atol=1e-4, what="preds")
    assert torch.equal(a, gd["anchors"]) and torch.equal(s, gd["strides"])
    loss, ld = YoloDFLQFLoss(num_classes=80)(preds, [l3["gt0"], l3["gt1"]], a, s)
    close(ld["total_loss"], gd["total"], rtol=1e-4), close(ld["box_loss"], gd["box"], rtol=1e-4)
    close(ld["cls_loss"], gd["cls"], rtol=1e-4)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    for k, v in gd.items():
        if k.startswith("grad:"):
            close(grads[k[5:]], v, rtol=2e-3, atol=1e-3 * float(v.abs().max()) + 1e-9, what=k)
    keys = [str(k) for k in gd["gradnorms_keys"]]
    mine = torch.tensor([float(grads[k].double().norm()) for k in keys])
    close(mine, torch.as_tensor(gd["gradnorms"]), rtol=5e-3, atol=1e-6, what="grad norms")
    sd = model.state_dict()
    close(sd["net.p1.0.norm.running_mean"], gd["rm:net.p1.0"], rtol=1e-4, atol=1e-6)
    close(sd["head.cls.2.3.norm.running_var"], gd["rv:head.cls.2.3"], rtol=1e-4, atol=1e-6)
    assert int(sd["net.p1.0.norm.num_batches_tracked"]) == 1


def test_model_eval_fuse_inference_wiring():
    from src.model.model_builder import Model
    gd = load_golden("model_n320_eval")
    model = Model(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256], num_classes=80)
    det_fill_(model.state_dict(), int(gd["seed"]))
    img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(50))
    model.eval()
    with torch.no_grad():
        preds, _, _ = model(img)
        close(preds[:, :, ::7], gd["preds_stride7"], rtol=1e-3, atol=1e-4)
        dets = model.inference(img, conf_thres=0.0, iou_thres=0.45)
        model.fuse()
        pf, _, _ = model(img)
        close(pf[:, :, ::7], gd["fused_stride7"], rtol=1e-3, atol=1e-4)
    for i, dt in enumerate(dets):
        want = gd[f"det:{i}"]
        want = want.reshape(-1, 6) if isinstance(want, torch.Tensor) else torch.zeros(0, 6)
        assert dt.shape == want.shape
        close(dt, want, rtol=1e-3, atol=1e-2)


def test_product_refuses_cpu_tensors_without_emulation(monkeypatch):
    monkeypatch.undo()                    # real leaves back
    from src.hipops import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv_fwd(ops.new_nhwc(1, 8, 4, 4, torch.float32, "cpu"), torch.zeros(8 * 32), None, 8, 1, 1)


def test_validation_host_wiring():
    """decode_predictions / DetectionMetrics host logic (packing, offsets, counter layout, compute) against the
    reference's goldens, with the two HIP leaves emulated"""
    from src.training.metrics import DetectionMetrics
    from src.training.train_model import decode_predictions
    gd = load_golden("decode_val")
    out = decode_predictions(gd["preds"], gd["anchors"], gd["strides"], conf_threshold=0.6, top_k=10)
    for i in range(2):
        close(out[i], gd[f"out{i}"].reshape(-1, 5), what=f"decode{i}")
    gm = load_golden("metrics")
    po, go = gm["pred_off"], gm["gt_off"]
    m = DetectionMetrics(int(gm["num_classes"]), 0.5)
    for i in range(len(po) - 1):
        m.update(gm["pred"][po[i]:po[i + 1]], gm["gt"][go[i]:go[i + 1]])
    res = m.compute()
    for k, v in zip(gm["thr0.5:compute_keys"], gm["thr0.5:compute_vals"].tolist()):
        assert abs(res[str(k)] - v) <= 1e-12 + 1e-7 * abs(v), (k, res[str(k)], v)
    assert m.class_tp.tolist() == gm["thr0.5:class_tp"].long().tolist()


def test_checkpoints_load_under_any_wrapper_prefix(tmp_path):
    """A reference-style DDP checkpoint ("module."-prefixed keys, which the reference's own Model.load_weights
    rejects) loads into a bare Model, and load_checkpoint resumes a model + optimizer from a bare-key checkpoint"""
    from src.model.model_builder import Model
    from src.training.utils_train import canonical_state_dict, load_checkpoint, save_checkpoint
    cfg = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
    torch.manual_seed(1)
    a, b = Model(**cfg, num_classes=8), Model(**cfg, num_classes=8)
    ddp_style = {"module." + k: v.clone() for k, v in a.state_dict().items()}
    assert set(canonical_state_dict(ddp_style)) == set(a.state_dict())
    p = tmp_path / "ref_ddp.pth"
    torch.save({"epoch": 7, "model_state": ddp_style, "val_loss": 1.0}, p)
    b.load_weights(str(p))
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(va, vb), k
    opt_a = torch.optim.AdamW(a.parameters(), lr=1e-3)
    for q in a.parameters():
        q.grad = torch.ones_like(q) * 1e-3
    opt_a.step()
    save_checkpoint(a, opt_a, 3, 0.5, checkpoint_dir=str(tmp_path))
    c = Model(**cfg, num_classes=8)
    opt_c = torch.optim.AdamW(c.parameters(), lr=1e-3)
    assert load_checkpoint(c, opt_c, str(tmp_path / "model_epoch_3.pth")) == 3
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), c.state_dict().values()))
    sa, sc = opt_a.state_dict()["state"], opt_c.state_dict()["state"]
    assert sa.keys() == sc.keys() and all(torch.equal(sa[k]["exp_avg"], sc[k]["exp_avg"]) for k in sa)


def test_device_prefetcher_is_transparent_without_a_gpu():
    """DevicePrefetcher on the CPU: the loader's own items, its length and its attributes (sampler / batch_size) -- the epoch
    loops wrap every loader in it, so the CPU launch (BASELINE config 1) must see no difference."""
    from src.data.data_loader import DevicePrefetcher

    class L(list):
        sampler, batch_size = "S", 2
    items = L([(torch.zeros(2, 3, 4, 4), [{"boxes": torch.zeros(1, 5)}]), (torch.ones(2, 3, 4, 4), [{"boxes": torch.ones(2, 5)}])])
    pf = DevicePrefetcher(items, "cpu")
    got = list(pf)
    assert len(pf) == 2 and got[0][0] is items[0][0] and got[1][1] is items[1][1]
    assert pf.sampler == "S" and pf.batch_size == 2
    with pytest.raises(AttributeError):
        pf.no_such_attribute
