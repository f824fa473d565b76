"""-m gpu: size-independent properties at BASELINE.json's full sizes (preset s layers at 640x640 / 32 images, the
config-5 NMS tensor), where the CPU oracle would take too long: linearity of the conv kernels in each operand,
consistency between the three conv kernels (adjoint identity <dgrad(g), x> = <g, fwd(x)> = <wgrad(x, g), w>),
BatchNorm normalisation invariants, idempotence and sortedness of NMS."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from src.hipops import ops
    return ops


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


FULL = [(32, 64, 160, 160, 64, 3, 1), (32, 128, 160, 160, 128, 3, 2), (32, 96, 160, 160, 128, 1, 1), (32, 256, 40, 40, 256, 3, 1),
        (32, 512, 20, 20, 512, 1, 1)]


@pytest.mark.parametrize("n,cin,h,w,cout,k,s", FULL)
def test_conv_kernels_are_linear_and_mutually_adjoint_at_full_size(n, cin, h, w, cout, k, s):
    o = _ops()
    g = torch.Generator().manual_seed(cin * 7 + cout)
    bf = torch.bfloat16
    # values with few mantissa bits: products and the sums of a few thousand of them are exact in fp32, and the
    # bf16 rounding of the conv output is the only inexact step
    x1 = nhwc((torch.randint(-4, 5, (n, cin, h, w), generator=g) / 4).to(bf).cuda())
    x2 = nhwc((torch.randint(-4, 5, (n, cin, h, w), generator=g) / 4).to(bf).cuda())
    wt = (torch.randint(-4, 5, (cout, cin, k, k), generator=g) / 8).cuda()
    wp, wb = o.pack_weights(wt, k, s, 0, bf), o.pack_weights(wt, k, s, 1, bf)
    y1, y2, y12 = (o.conv_fwd(t, wp, None, cout, k, s).float() for t in (x1, x2, nhwc(x1 + x2)))
    scale = float(y12.abs().max())
    assert float((y12 - (y1 + y2)).abs().max()) <= 2 ** -7 * scale          # linear in x up to bf16 output rounding
    gy = nhwc((torch.randint(-2, 3, y1.shape, generator=g) / 2).to(bf).cuda())
    dx = o.conv_dgrad(gy, wb, cin, h, w, k, s).float()
    dw = o.conv_wgrad(x1, gy, k, s, torch.float32)
    a = float((gy.float() * y1).sum())                                       # <g, fwd(x)>  (y1 rounded to bf16)
    b = float((dx * x1.float()).sum())                                       # <dgrad(g), x>
    c = float((dw * wt).sum())                                               # <wgrad(x, g), w>  (fp32 accumulate, exact-ish)
    # the three inner products cancel heavily (random signs): compare against the absolute mass of the sum; y1 and
    # dx carry one bf16 rounding per element (unbiased, so it averages out over millions of terms)
    ref = float((gy.float().abs() * y1.abs()).sum())
    assert abs(b - c) / ref < 2e-4 and abs(a - c) / ref < 2e-4, (a, b, c, ref)


@pytest.mark.parametrize("n,c,h,w", [(32, 128, 160, 160), (32, 32, 320, 320), (32, 512, 20, 20)])
def test_batchnorm_training_output_is_normalised_at_full_size(n, c, h, w):
    """gamma = 1, beta = 0, identity activation: per-channel mean 0 / variance 1 of the output; the backward of a
    constant upstream gradient is (numerically) zero, and dgamma = sum(dout * yhat), dbeta = sum(dout)."""
    o = _ops()
    g = torch.Generator().manual_seed(c)
    y = nhwc((torch.randn(n, c, h, w, generator=g) * 3 + 1.5).to(torch.bfloat16).cuda())
    acc = o.bn_acc_new(c, y.device)
    o.bn_stats_acc(y, acc)
    ones, zeros = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    mean, invstd, scale, shift = o.bn_finalize_acc(acc, n * h * w, ones, zeros, zeros.clone(), ones.clone(), 0.03, 1e-3)
    out = o.bn_act_fwd(y, scale, shift, 0).float()
    m, v = out.mean((0, 2, 3)), out.var((0, 2, 3), unbiased=False)
    assert float(m.abs().max()) < 2e-2 and float((v - 1).abs().max()) < 2e-2
    dout = nhwc(torch.ones_like(y))
    dy, dgamma, dbeta = o.bn_act_bwd(dout, y, scale, shift, mean, invstd, ones, 0)
    assert float(dy.float().abs().max()) < 5e-2                              # d/dy of sum(BN(y)) = 0
    assert torch.allclose(dbeta, torch.full_like(dbeta, float(n * h * w)), rtol=1e-5)
    assert float(dgamma.abs().max()) < 1e-2 * n * h * w                      # sum(yhat) ~ 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_nms_is_idempotent_and_sorted_at_config5_size(dtype):
    """Config 5: (2, 84, 33600).  Kept rows are sorted by confidence, at most max_det, pairwise IoU within a class
    below the threshold, and running NMS again on the kept boxes keeps all of them."""
    from src.utils.model_utils import non_max_suppression
    g = torch.Generator().manual_seed(5)
    bs, nc, M = 2, 80, 33600
    cxcy = torch.rand(bs, 2, M, generator=g) * 1280
    wh = torch.rand(bs, 2, M, generator=g) * 200 + 10
    cls = torch.rand(bs, nc, M, generator=g) * 0.2
    hot = torch.randint(0, nc, (bs, M), generator=g)
    cls.scatter_(1, hot.unsqueeze(1), torch.rand(bs, 1, M, generator=g) * 0.7 + 0.3)
    pred = torch.cat([cxcy, wh, cls], 1).to(dtype).cuda()
    dets = non_max_suppression(pred, conf_thres=0.5, iou_thres=0.45)
    for d in dets:
        assert 0 < d.shape[0] <= 300 and d.shape[1] == 6
        conf = d[:, 4].float()
        assert bool((conf[:-1] >= conf[1:]).all())
        b, c = d[:, :4].float(), d[:, 5]
        x1 = torch.max(b[:, None, 0], b[None, :, 0]); y1 = torch.max(b[:, None, 1], b[None, :, 1])
        x2 = torch.min(b[:, None, 2], b[None, :, 2]); y2 = torch.min(b[:, None, 3], b[None, :, 3])
        inter = (x2 - x1).clamp(0) * (y2 - y1).clamp(0)
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        iou = inter / (area[:, None] + area[None, :] - inter)
        same = (c[:, None] == c[None, :]) & ~torch.eye(len(c), dtype=torch.bool, device=c.device)
        assert float((iou * same).max()) <= 0.45 + 2e-3
        # idempotence: feed the kept boxes back (as xywh + one-hot class scores)
        k = d.shape[0]
        xywh = torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 0)
        sc = torch.zeros(nc, k, device=d.device)
        sc[c.long(), torch.arange(k, device=d.device)] = conf
        again = non_max_suppression(torch.cat([xywh, sc], 0).unsqueeze(0).to(dtype), conf_thres=0.5, iou_thres=0.45)[0]
        assert again.shape[0] == k
