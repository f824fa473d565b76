"""Plain-torch CPU stand-ins for every leaf of src.hipops.ops (TEST INFRASTRUCTURE).

Two uses: (1) `-m "not gpu"` tests monkeypatch the leaves with these to exercise the host wiring
(autograd Functions, modules, state dicts) where no GPU exists; (2) `-m gpu` tests run the real HIP
leaf on device tensors and compare with the stand-in on CPU copies, kernel by kernel.
The product never imports this file."""
import torch
import torch.nn.functional as F

from oracle import loss as oloss
from oracle import postproc as opost
from oracle.params import ParamStore

from src.hipops import ops as real


def _nhwc(t):
    n, c, h, w = t.shape
    out = real.new_nhwc(n, c, h, w, t.dtype, t.device)
    out.copy_(t)
    return out


def to_nhwc(x, dtype):
    if real.is_nhwc(x) and x.dtype == dtype:
        return x
    return _nhwc(x.to(dtype))


def head_pack(x, preds, c_off, m_off):
    n, c, h, w = x.shape
    preds[:, c_off:c_off + c, m_off:m_off + h * w] = x.reshape(n, c, h * w).to(preds.dtype)


def head_unpack(dpreds, c_off, c, m_off, h, w):
    n = dpreds.shape[0]
    return _nhwc(dpreds[:, c_off:c_off + c, m_off:m_off + h * w].reshape(n, c, h, w))


def head_group(branches, preds, c_offs, m_offs, pack):
    for b, c_off, m_off in zip(branches, c_offs, m_offs):
        n, c, h, w = b.shape
        if pack:
            head_pack(b, preds, c_off, m_off)
        else:
            b.copy_(preds[:, c_off:c_off + c, m_off:m_off + h * w].reshape(n, c, h, w))


def copy_channels(src, dst, accumulate=False):
    if accumulate:
        dst.copy_((dst.float() + src.float()).to(dst.dtype))
    else:
        dst.copy_(src)


def add_n(xs, out=None):
    acc = xs[0].float()
    for x in xs[1:]:
        acc = acc + x.float()
    return _into(out, _nhwc(acc.to(xs[0].dtype)))


def bucket_copy(plan, scale=1.0):
    for s_, d_ in zip(plan["srcs"], plan["dsts"]):
        d_.copy_((s_.float() * scale).to(d_.dtype))


def zero_(t):
    return t.zero_()


def fill_(t, v):
    return t.fill_(v)


def pack_weights(w, k, stride, mode, dtype):
    return w.detach().to(dtype)           # stand-in handle: the OIHW weights rounded to the compute dtype


def _into(out, val):
    """Producers may be handed their destination (a channel slice of a concat buffer)."""
    if out is None:
        return val
    out.data.copy_(val)       # like the kernels: a raw write, invisible to autograd's version counters
    return out


def conv_fwd(x, wp, bias, cout, k, stride, stats_acc=None, out=None):
    y = _into(out, _nhwc(F.conv2d(x.float(), wp.float(), bias, stride, k // 2).to(x.dtype)))
    if stats_acc is not None:
        bn_stats_acc(y, stats_acc)
    return y


def bn_stats_acc(y, acc):
    c = y.shape[1]
    a = acc.view(real.BN_REPL, 2, c)
    yf = y.float()
    a[0, 0] += yf.sum((0, 2, 3))
    a[0, 1] += (yf * yf).sum((0, 2, 3))


def bn_finalize_acc(acc, count, gamma, beta, running_mean, running_var, momentum, eps):
    c = gamma.numel()
    s = acc.view(real.BN_REPL, 2, c).sum(0)
    mean = s[0] / count
    var = (s[1] / count - mean * mean).clamp_min(0)
    invstd = 1.0 / torch.sqrt(var + eps)
    running_mean.mul_(1 - momentum).add_(momentum * mean)
    running_var.mul_(1 - momentum).add_(momentum * var * (count / max(count - 1, 1)))
    scale = gamma * invstd
    return mean, invstd, scale, beta - mean * scale


def bn_act_fwd_train(y, acc, gamma, beta, running_mean, running_var, momentum, eps, act, res=None, out=None):
    c = y.shape[1]
    cnt = y.numel() // c
    s = acc.view(real.BN_REPL, 2, c).sum(0)
    mean = s[0] / cnt
    var = (s[1] / cnt - mean * mean).clamp_min(0)
    invstd = 1.0 / torch.sqrt(var + eps)
    running_mean.mul_(1 - momentum).add_(momentum * mean)
    running_var.mul_(1 - momentum).add_(momentum * var * (cnt / max(cnt - 1, 1)))
    scale = gamma * invstd
    shift = beta - mean * scale
    return bn_act_fwd(y, scale, shift, act, res, out), mean, invstd, scale, shift


def bn_act_bwd_train(dout, y, scale, shift, mean, invstd, gamma, act, acc):
    return bn_act_bwd(dout, y, scale, shift, mean, invstd, gamma, act)


def conv_dgrad(dy, wb, cin, h, w, k, stride, acc_into=None, acc2=None):
    n = dy.shape[0]
    dx = torch.nn.grad.conv2d_input((n, cin, h, w), wb.float(), dy.float(), stride, k // 2)
    if acc2 is not None:
        dx = dx + acc2.float()
    if acc_into is not None:
        acc_into.data.copy_((acc_into.float() + dx).to(acc_into.dtype))
        return acc_into
    return _nhwc(dx.to(dy.dtype))


def conv_wgrad(x, dy, k, stride, w_dtype):
    dw = torch.nn.grad.conv2d_weight(x.float(), (dy.shape[1], x.shape[1], k, k), dy.float(), stride, k // 2)
    return dw.to(w_dtype)


def stem_im2col(img, dtype):
    n, c, h, w = img.shape
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    cols = F.unfold(img.float(), 3, padding=1, stride=2).view(n, 27, oh, ow)      # k = ci*9 + kh*3 + kw
    return _nhwc(F.pad(cols, (0, 0, 0, 0, 0, 5)).to(dtype))


def stem_pack_weights(w, dtype):
    return F.pad(w.detach().reshape(w.shape[0], 27), (0, 5)).reshape(w.shape[0], 32, 1, 1).to(dtype)


def stem_unpack_wgrad(dw32, w_dtype, out=None):
    return dw32.reshape(dw32.shape[0], 32)[:, :27].reshape(-1, 3, 3, 3).to(w_dtype)


def dw_fwd(x, w9, stats_acc=None):
    c = x.shape[1]
    y = _nhwc(F.conv2d(x.float(), w9.view(c, 1, 3, 3), None, 1, 1, 1, c).to(x.dtype))
    if stats_acc is not None:
        bn_stats_acc(y, stats_acc)
    return y


def _acc(acc_into, val, dtype):
    if acc_into is None:
        return _nhwc(val.to(dtype))
    acc_into.data.copy_((acc_into.float() + val).to(acc_into.dtype))
    return acc_into


def dw_dgrad(dy, w9, acc_into=None):
    c = dy.shape[1]
    dx = torch.nn.grad.conv2d_input(tuple(dy.shape), w9.view(c, 1, 3, 3), dy.float(), 1, 1, 1, c)
    return _acc(acc_into, dx, dy.dtype)


def dw_wgrad(x, dy):
    c = x.shape[1]
    return torch.nn.grad.conv2d_weight(x.float(), (c, 1, 3, 3), dy.float(), 1, 1, 1, c)


def bn_train_stats(y, gamma, beta, running_mean, running_var, momentum, eps):
    yf = y.float()
    cnt = yf.numel() // yf.shape[1]
    mean = yf.mean((0, 2, 3))
    var = yf.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + eps)
    running_mean.mul_(1 - momentum).add_(momentum * mean)
    running_var.mul_(1 - momentum).add_(momentum * var * (cnt / max(cnt - 1, 1)))
    scale = gamma * invstd
    return mean, invstd, scale, beta - mean * scale


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    scale = gamma / torch.sqrt(running_var + eps)
    return scale, beta - running_mean * scale


def _act(z, act):
    return F.silu(z) if act == 1 else z


def _act_grad(z, act):
    if act == 0:
        return torch.ones_like(z)
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


def bn_act_fwd(y, scale, shift, act, res=None, out=None):
    z = _act(y.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), act)
    if res is not None:
        z = z + res.float()
    return _into(out, _nhwc(z.to(y.dtype)))


def bn_act_bwd(dout, y, scale, shift, mean, invstd, gamma, act):
    v = lambda t: t.view(1, -1, 1, 1)
    yf = y.float()
    dz = dout.float() * _act_grad(yf * v(scale) + v(shift), act)
    yh = (yf - v(mean)) * v(invstd)
    m = yf.numel() // yf.shape[1]
    dbeta = dz.sum((0, 2, 3))
    dgamma = (dz * yh).sum((0, 2, 3))
    dy = v(gamma.float() * invstd) * (dz - v(dbeta) / m - yh * v(dgamma) / m)
    return _nhwc(dy.to(y.dtype)), dgamma.to(gamma.dtype), dbeta.to(gamma.dtype)     # like the kernels: the parameter's dtype


def bn_act_bwd_eval(dout, y, scale, shift, act):
    v = lambda t: t.view(1, -1, 1, 1)
    dz = dout.float() * _act_grad(y.float() * v(scale) + v(shift), act)
    return _nhwc((v(scale) * dz).to(y.dtype))


def channel_sum(x):
    return x.float().sum((0, 2, 3))


def maxpool5_fwd(x, out=None):
    o, idx = F.max_pool2d(x.float(), 5, 1, 2, return_indices=True)
    return _into(out, _nhwc(o.to(x.dtype))), idx


def maxpool5_bwd(dout, idx, acc_into=None):
    n, c, h, w = dout.shape
    dx = torch.zeros(n, c, h * w)
    dx.scatter_add_(2, idx.reshape(n, c, -1), dout.float().reshape(n, c, -1))
    return _acc(acc_into, dx.view(n, c, h, w), dout.dtype)


def upsample2x_fwd(x, out=None):
    return _into(out, _nhwc(F.interpolate(x.float(), scale_factor=2.0, mode="nearest").to(x.dtype)))


def upsample2x_bwd(dout, acc_into=None):
    n, c, oh, ow = dout.shape
    return _acc(acc_into, dout.float().view(n, c, oh // 2, 2, ow // 2, 2).sum((3, 5)), dout.dtype)


def _split_qkv(qkv, heads, dk, dh):
    n, cq, h, w = qkv.shape
    t = qkv.float().reshape(n, heads, 2 * dk + dh, h * w)
    return t[:, :, :dk], t[:, :, dk:2 * dk], t[:, :, 2 * dk:]


def attn_fwd(qkv, heads, dk, dh, scale):
    n, cq, h, w = qkv.shape
    q, k, v = _split_qkv(qkv, heads, dk, dh)
    s = (q.transpose(-2, -1) @ k) * scale
    lse = torch.logsumexp(s, -1)
    o = v @ s.softmax(-1).transpose(-2, -1)
    return (_nhwc(o.reshape(n, heads * dh, h, w).to(qkv.dtype)), _nhwc(v.reshape(n, heads * dh, h, w).to(qkv.dtype)),
            lse.reshape(-1))


def attn_bwd(qkv, o, d_o, d_vp, lse, heads, dk, dh, scale):
    n, cq, h, w = qkv.shape
    x = qkv.detach().float().clone().requires_grad_(True)
    with torch.enable_grad():
        q, k, v = _split_qkv(x, heads, dk, dh)
        att = ((q.transpose(-2, -1) @ k) * scale).softmax(-1)
        oo = (v @ att.transpose(-2, -1)).reshape(n, heads * dh, h, w)
        vv = v.reshape(n, heads * dh, h, w)
        outs, grads = [oo], [d_o.float()]
        if d_vp is not None:
            outs.append(vv)
            grads.append(d_vp.float())
        (g,) = torch.autograd.grad(outs, x, grads)
    return _nhwc(g.to(qkv.dtype))


def loss_fwd_bwd(preds, anchors, strides, gt, gt_off, gt_img, n_gt, nc, lambda_dfl, lambda_cls, want_grad, grad_scale=None):
    offs = gt_off.tolist()
    gts = [gt[offs[i]:offs[i + 1]].reshape(-1, 5) for i in range(preds.shape[0])]
    p = preds.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        tot, dfl, cls = oloss.dfl_qfl_loss(p, gts, anchors.to(preds.dtype), strides.to(preds.dtype), nc,
                                           lambda_cls=lambda_cls, lambda_dfl=lambda_dfl)
        dp = torch.autograd.grad(tot, p)[0] if want_grad else None
    return torch.stack([tot.detach(), dfl.detach(), cls.detach()]).float(), dp, None


def _vjp(fn, args, g):
    leaves = [a.detach().clone().requires_grad_(True) for a in args]
    with torch.enable_grad():
        out = fn(*leaves)
    return torch.autograd.grad(out, leaves, g)


def bbox_iou(b1, b2, g=None):
    return oloss.quirk_iou(b1, b2) if g is None else _vjp(oloss.quirk_iou, (b1, b2), g)


def qfl(pred, target, beta, g=None):
    f = lambda p, t: oloss.qfl_sum(p, t, beta)
    return f(pred, target) if g is None else _vjp(f, (pred, target), g)


def dfl_loss(pred_dist, target_val, g=None):
    return oloss.dfl_side(pred_dist, target_val) if g is None else _vjp(oloss.dfl_side, (pred_dist, target_val), g)


def scale_inplace(x, scale_dev):
    return x.mul_(scale_dev.to(x.dtype))


def image_prep(src_u8, recs, size, jitter, dtype, mean, std):
    from oracle.image_prep import transform_image
    outs = []
    for off, h, w, flip, order, fac in recs:
        img = src_u8[off:off + h * w * 3].reshape(h, w, 3).numpy()
        outs.append(transform_image(img, size, flip, tuple(order) if jitter else (), fac, mean, std))
    return torch.stack(outs).to(dtype)


def head_decode(preds, anchors, strides, nc):
    return opost.inference_decode(ParamStore(), preds.float(), anchors.float(), strides.float(), nc).to(preds.dtype)


def dfl_expect(x):
    from oracle.blocks import dfl
    return dfl(ParamStore(), x.float()).to(x.dtype)


def nms(y, nc, conf_thres, iou_thres, classes, agnostic, multi_label, max_det):
    res = opost.non_max_suppression(y, conf_thres, iou_thres, classes, agnostic, multi_label, (), max_det, nc)
    bs = y.shape[0]
    rows = torch.zeros(bs, max_det, 6)
    counts = torch.zeros(bs, dtype=torch.int32)
    for i, r in enumerate(res):
        rows[i, :r.shape[0]] = r
        counts[i] = r.shape[0]
    return rows, counts, torch.zeros(1, dtype=torch.int32)


def val_select(y, nc, conf_threshold, top_k):
    """rows [N][top_k][6] = cx, cy, w, h, cls, score; counts -- train_model.py:102-139 on the decoded tensor"""
    n, _, m = y.shape
    rows = torch.zeros(n, top_k, 6)
    counts = torch.zeros(n, dtype=torch.int32)
    for b in range(n):
        sc, ci = y[b, 4:].transpose(0, 1).sigmoid().max(1)
        keep = sc >= conf_threshold
        bb, ss, cc = y[b, :4].transpose(0, 1)[keep], sc[keep], ci[keep]
        if ss.numel() > top_k:
            top = opost.stable_desc_order(ss.float())[:top_k]
            bb, ss, cc = bb[top], ss[top], cc[top]
        k = ss.numel()
        rows[b, :k, :4], rows[b, :k, 4], rows[b, :k, 5] = bb.float(), cc.float(), ss.float()
        counts[b] = k
    return rows, counts


def val_match(rows, count, gt, gt_off, iou_threshold, nc, skip_empty_gt, counters, status):
    mc = opost.MetricCounters(nc, iou_threshold)
    for b in range(rows.shape[0]):
        t = gt[int(gt_off[b]):int(gt_off[b + 1])]
        if skip_empty_gt and t.shape[0] == 0:
            continue
        mc.update(rows[b, :int(count[b]), :5], t)
    counters += torch.tensor(mc.scalars() + mc.class_tp.tolist() + mc.class_fp.tolist() + mc.class_fn.tolist()
                             + mc.class_gt.tolist(), dtype=torch.int64)


LEAVES = ["to_nhwc", "head_pack", "head_unpack", "head_group", "copy_channels", "add_n", "bucket_copy", "zero_", "fill_", "pack_weights", "conv_fwd",
          "stem_im2col", "stem_pack_weights", "stem_unpack_wgrad", "bn_stats_acc", "bn_finalize_acc", "bn_act_fwd_train", "bn_act_bwd_train",
          "conv_dgrad", "conv_wgrad", "dw_fwd", "dw_dgrad", "dw_wgrad", "bn_train_stats", "bn_eval_coeffs",
          "bn_act_fwd", "bn_act_bwd", "bn_act_bwd_eval", "channel_sum", "maxpool5_fwd", "maxpool5_bwd",
          "upsample2x_fwd", "upsample2x_bwd", "attn_fwd", "attn_bwd", "loss_fwd_bwd", "image_prep", "bbox_iou", "qfl", "dfl_loss", "scale_inplace", "head_decode",
          "dfl_expect", "nms", "val_select", "val_match"]


def install(monkeypatch):
    """Swap every HIP leaf for its stand-in (CPU wiring tests only)."""
    import sys
    me = sys.modules[__name__]
    for name in LEAVES:
        monkeypatch.setattr(real, name, getattr(me, name))


def install_plain():
    """Same swap without pytest (spawned worker processes)."""
    import sys
    me = sys.modules[__name__]
    for name in LEAVES:
        setattr(real, name, getattr(me, name))
