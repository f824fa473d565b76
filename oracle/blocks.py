"""Functional fp32 restatement of the reference network (test infrastructure).

Every function takes ``ps`` (a ParamStore / plain dict keyed by the reference's
state-dict names), a key prefix and NCHW tensors, and uses stock torch CPU ops.
Citations are into the reference checkout (src/model/...).
"""
import torch
import torch.nn.functional as F

from .params import ParamStore

BN_EPS = 1e-3        # src/model/model_blocks.py:28
BN_MOMENTUM = 0.03   # src/model/model_blocks.py:28


class Ctx:
    """Forward-mode flags: BN batch statistics (train) vs running statistics (eval)."""

    def __init__(self, training: bool = True, fused: bool = False):
        self.training = training
        self.fused = fused


def _need(ps, key, shape):
    return ps.want(key, shape) if isinstance(ps, ParamStore) else ps[key]


def conv_block(ps, pre, x, cin, cout, act, ctx, k=1, s=1, p=0, g=1):
    """Conv = conv2d(bias=False) -> BatchNorm2d(eps 1e-3, momentum 0.03) -> act.
    src/model/model_blocks.py:26-37; fused (BN folded, bias) form :36-37."""
    w = _need(ps, pre + ".conv.weight", (cout, cin // g, k, k))
    if ctx.fused:
        b = _need(ps, pre + ".conv.bias", (cout,))
        y = F.conv2d(x, w, b, s, p, 1, g)
    else:
        y = F.conv2d(x, w, None, s, p, 1, g)
        gamma = _need(ps, pre + ".norm.weight", (cout,))
        beta = _need(ps, pre + ".norm.bias", (cout,))
        rm = _need(ps, pre + ".norm.running_mean", (cout,))
        rv = _need(ps, pre + ".norm.running_var", (cout,))
        if isinstance(ps, ParamStore):
            nb = ps.want(pre + ".norm.num_batches_tracked", ())
        else:
            nb = ps.get(pre + ".norm.num_batches_tracked")
        if ctx.training and nb is not None:
            nb += 1
        y = F.batch_norm(y, rm, rv, gamma, beta, ctx.training, BN_MOMENTUM, BN_EPS)
    return F.silu(y) if act else y


def residual(ps, pre, x, ch, ctx, e=0.5):
    """x + Conv3x3(Conv3x3(x)); src/model/model_blocks.py:56-62."""
    mid = int(ch * e)
    y = conv_block(ps, pre + ".conv1", x, ch, mid, True, ctx, k=3, p=1)
    y = conv_block(ps, pre + ".conv2", y, mid, ch, True, ctx, k=3, p=1)
    return x + y


def c3k(ps, pre, x, cin, cout, ctx):
    """src/model/model_blocks.py:80-92."""
    h = cout // 2
    a = conv_block(ps, pre + ".conv1", x, cin, h, True, ctx)
    a = residual(ps, pre + ".res_m.0", a, h, ctx, e=1.0)
    a = residual(ps, pre + ".res_m.1", a, h, ctx, e=1.0)
    b = conv_block(ps, pre + ".conv2", x, cin, h, True, ctx)
    return conv_block(ps, pre + ".conv3", torch.cat((a, b), 1), 2 * h, cout, True, ctx)


def c3k2(ps, pre, x, cin, cout, n, csp, r, ctx):
    """src/model/model_blocks.py:112-125."""
    h = cout // r
    parts = list(conv_block(ps, pre + ".conv1", x, cin, 2 * h, True, ctx).chunk(2, 1))
    for i in range(n):
        if csp:
            parts.append(c3k(ps, f"{pre}.res_m.{i}", parts[-1], h, h, ctx))
        else:
            parts.append(residual(ps, f"{pre}.res_m.{i}", parts[-1], h, ctx))
    return conv_block(ps, pre + ".conv2", torch.cat(parts, 1), (2 + n) * h, cout, True, ctx)


def sppf(ps, pre, x, c1, c2, ctx, k=5):
    """src/model/model_blocks.py:147-156."""
    h = c1 // 2
    x = conv_block(ps, pre + ".cv1", x, c1, h, True, ctx)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return conv_block(ps, pre + ".cv2", torch.cat((x, y1, y2, y3), 1), 4 * h, c2, True, ctx)


def attention(ps, pre, x, ch, heads, ctx):
    """src/model/model_blocks.py:176-198."""
    dh = ch // heads
    dk = dh // 2
    b, c, hh, ww = x.shape
    qkv = conv_block(ps, pre + ".qkv", x, ch, ch + dk * heads * 2, False, ctx)
    qkv = qkv.view(b, heads, 2 * dk + dh, hh * ww)
    q, k_, v = qkv.split([dk, dk, dh], dim=2)
    att = ((q.transpose(-2, -1) @ k_) * dk ** -0.5).softmax(dim=-1)
    o = (v @ att.transpose(-2, -1)).view(b, c, hh, ww)
    o = o + conv_block(ps, pre + ".conv1", v.reshape(b, c, hh, ww), ch, ch, False, ctx, k=3, p=1, g=ch)
    return conv_block(ps, pre + ".conv2", o, ch, ch, False, ctx)


def psablock(ps, pre, x, ch, heads, ctx):
    """src/model/model_blocks.py:216-224."""
    x = x + attention(ps, pre + ".conv1", x, ch, heads, ctx)
    y = conv_block(ps, pre + ".conv2.0", x, ch, 2 * ch, True, ctx)
    y = conv_block(ps, pre + ".conv2.1", y, 2 * ch, ch, False, ctx)
    return x + y


def psa(ps, pre, x, ch, n, ctx):
    """src/model/model_blocks.py:243-252."""
    h = ch // 2
    a, b = conv_block(ps, pre + ".conv1", x, ch, 2 * h, True, ctx).chunk(2, 1)
    for i in range(n):
        b = psablock(ps, f"{pre}.res_m.{i}", b, h, ch // 128, ctx)
    return conv_block(ps, pre + ".conv2", torch.cat((a, b), 1), 2 * h, ch, True, ctx)


def dfl(ps, x, c1=16):
    """(b, 4*c1, a) -> (b, 4, a): softmax over bins then dot with 0..c1-1.
    src/model/model_blocks.py:273-280."""
    w = _need(ps, "head.dfl.conv.weight", (1, c1, 1, 1))
    b, _, a = x.shape
    return F.conv2d(x.view(b, 4, c1, a).transpose(2, 1).softmax(1), w).view(b, 4, a)


def backbone(ps, x, width, depth, csp, ctx):
    """src/model/backbone.py:37-66."""
    w = width
    p1 = conv_block(ps, "net.p1.0", x, w[0], w[1], True, ctx, k=3, s=2, p=1)
    p2 = conv_block(ps, "net.p2.0", p1, w[1], w[2], True, ctx, k=3, s=2, p=1)
    p2 = c3k2(ps, "net.p2.1", p2, w[2], w[3], depth[0], csp[0], 4, ctx)
    p3 = conv_block(ps, "net.p3.0", p2, w[3], w[3], True, ctx, k=3, s=2, p=1)
    p3 = c3k2(ps, "net.p3.1", p3, w[3], w[4], depth[1], csp[0], 4, ctx)
    p4 = conv_block(ps, "net.p4.0", p3, w[4], w[4], True, ctx, k=3, s=2, p=1)
    p4 = c3k2(ps, "net.p4.1", p4, w[4], w[4], depth[2], csp[1], 2, ctx)
    p5 = conv_block(ps, "net.p5.0", p4, w[4], w[5], True, ctx, k=3, s=2, p=1)
    p5 = c3k2(ps, "net.p5.1", p5, w[5], w[5], depth[3], csp[1], 2, ctx)
    p5 = sppf(ps, "net.p5.2", p5, w[5], w[5], ctx)
    p5 = psa(ps, "net.p5.3", p5, w[5], depth[4], ctx)
    return p3, p4, p5


def neck(ps, feats, width, depth, csp, ctx):
    """src/model/neck.py:31-45 (nearest x2 upsample, concat, C3K2)."""
    w = width
    p3, p4, p5 = feats
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")
    p4 = c3k2(ps, "fpn.h1", torch.cat([up(p5), p4], 1), w[4] + w[5], w[4], depth[5], csp[0], 2, ctx)
    p3 = c3k2(ps, "fpn.h2", torch.cat([up(p4), p3], 1), w[4] + w[4], w[3], depth[5], csp[0], 2, ctx)
    d3 = conv_block(ps, "fpn.h3", p3, w[3], w[3], True, ctx, k=3, s=2, p=1)
    p4 = c3k2(ps, "fpn.h4", torch.cat([d3, p4], 1), w[3] + w[4], w[4], depth[5], csp[0], 2, ctx)
    d4 = conv_block(ps, "fpn.h5", p4, w[4], w[4], True, ctx, k=3, s=2, p=1)
    p5 = c3k2(ps, "fpn.h6", torch.cat([d4, p5], 1), w[4] + w[5], w[5], depth[5], csp[1], 2, ctx)
    return p3, p4, p5


def make_anchors(shapes_hw, strides, dtype=torch.float32, offset=0.5):
    """Grid centres (x fastest) and per-anchor stride; src/utils/model_utils.py:60-70."""
    pts, sts = [], []
    for (h, w), s in zip(shapes_hw, strides):
        sx = torch.arange(w, dtype=dtype) + offset
        sy = torch.arange(h, dtype=dtype) + offset
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((gx, gy), -1).view(-1, 2))
        sts.append(torch.full((h * w, 1), float(s), dtype=dtype))
    return torch.cat(pts), torch.cat(sts)


def head(ps, feats, filters, nc, strides, ctx):
    """src/model/head.py:41-62 (branches) and :86-121 (assembly)."""
    bw = max(64, filters[0] // 4)
    cw = max(80, filters[0], nc)
    outs = []
    for i, (x, f) in enumerate(zip(feats, filters)):
        pb = f"head.box.{i}"
        b = conv_block(ps, pb + ".0", x, f, bw, True, ctx, k=3, p=1)
        b = conv_block(ps, pb + ".1", b, bw, bw, True, ctx, k=3, p=1)
        b = F.conv2d(b, _need(ps, pb + ".2.weight", (64, bw, 1, 1)), _need(ps, pb + ".2.bias", (64,)))
        pc = f"head.cls.{i}"
        c = conv_block(ps, pc + ".0", x, f, f, True, ctx, k=3, p=1, g=f)
        c = conv_block(ps, pc + ".1", c, f, cw, True, ctx)
        c = conv_block(ps, pc + ".2", c, cw, cw, True, ctx, k=3, p=1, g=cw)
        c = conv_block(ps, pc + ".3", c, cw, cw, True, ctx)
        c = F.conv2d(c, _need(ps, pc + ".4.weight", (nc, cw, 1, 1)), _need(ps, pc + ".4.bias", (nc,)))
        outs.append(torch.cat((b, c), 1))
    anchors, st = make_anchors([o.shape[-2:] for o in outs], strides, dtype=outs[0].dtype)
    n = outs[0].shape[0]
    preds = torch.cat([o.reshape(n, 64 + nc, -1) for o in outs], 2)
    return preds, anchors.transpose(0, 1), st.transpose(0, 1)


MODEL_STRIDES = (8.0, 16.0, 32.0)  # src/model/model_builder.py:37-45 always yields these


def model_forward(ps, x, width, depth, csp, nc, training=True, fused=False):
    """Model.forward: (preds[N,64+nc,M], anchors[2,M], strides[1,M]); src/model/model_builder.py:47-50."""
    ctx = Ctx(training, fused)
    if isinstance(ps, ParamStore):
        ps.want("head.dfl.conv.weight", (1, 16, 1, 1))
    f = backbone(ps, x, width, depth, csp, ctx)
    f = neck(ps, f, width, depth, csp, ctx)
    return head(ps, list(f), (width[3], width[4], width[5]), nc, MODEL_STRIDES, ctx)


def fold_bn(w, gamma, beta, mean, var, eps=BN_EPS):
    """fuse_conv: W' = diag(g/sqrt(var+eps)) W, b' = beta - g*mean/sqrt(var+eps).
    src/utils/model_utils.py:110-116."""
    s = gamma / torch.sqrt(var + eps)
    return w * s.view(-1, 1, 1, 1), beta - mean * s


def fuse_state(ps):
    """State dict after Model.fuse(): every Conv's BN folded into .conv.weight/.conv.bias.
    src/model/model_builder.py:52-58."""
    out = {}
    for k, v in ps.items():
        if ".norm." in k:
            continue
        if k.endswith(".conv.weight") and (k[:-len("conv.weight")] + "norm.weight") in ps:
            pre = k[:-len("conv.weight")]
            w, b = fold_bn(v, ps[pre + "norm.weight"], ps[pre + "norm.bias"],
                           ps[pre + "norm.running_mean"], ps[pre + "norm.running_var"])
            out[k], out[pre + "conv.bias"] = w, b
        else:
            out[k] = v
    return out


PRESETS = {  # notebooks/03_training_experiements.ipynb:35-40, config.yaml:49-53
    "n": dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256]),
    "s": dict(csp=[False, True], depth=[1] * 6, width=[3, 32, 64, 128, 256, 512]),
    "m": dict(csp=[True, True], depth=[1] * 6, width=[3, 64, 128, 256, 512, 512]),
    "l": dict(csp=[True, True], depth=[2] * 6, width=[3, 64, 128, 256, 512, 512]),
    "x": dict(csp=[True, True], depth=[2] * 6, width=[3, 96, 192, 384, 768, 768]),
}
