"""Leaf ops: one function = one (or a fixed short chain of) C-ABI call(s) on torch-owned device memory.

Activations are logical (N, C, H, W) tensors whose memory is NHWC ("channels_last"), possibly a channel
slice of a wider buffer; `geom` extracts (N, C, H, W, ld).  PyTorch is used here only to allocate and
to carry pointers/streams.  Every function requires CUDA(HIP) tensors -- there is no CPU path.
"""
import os

import torch

from . import lib

ALGO = int(os.environ.get("YOLO_HIP_CONV_ALGO", "0"))   # 0 auto, 1 generic VALU kernels, 2 MFMA or error

_DT = {torch.float32: lib.F32, torch.bfloat16: lib.BF16, torch.float16: lib.F16}


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t}")


def _stream(t):
    if not t.is_cuda:
        raise RuntimeError("yolo_hip ops need tensors on the GPU: the product path has no CPU fallback "
                           "(the CPU oracle lives under oracle/ and is test infrastructure)")
    return _raw_stream(t.device.index)


# the current stream's handle without building a torch.cuda.Stream object per launch (4.7 us each, 2.5 ms of a 21.7 ms
# launch-by-launch step of preset s: tools/host_profile.py)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or \
    (lambda index: torch.cuda.current_stream(index).cuda_stream)


def _p(t):
    return 0 if t is None else t.data_ptr()


def new_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def _dest(out, n, c, h, w, dtype, device):
    """Destination of a producer: a fresh NHWC tensor, or the caller's view (typically a channel slice of a concat
    buffer, so the consumer's torch.cat costs no copy -- every kernel takes a row stride `ld`)."""
    if out is None:
        return new_nhwc(n, c, h, w, dtype, device)
    if tuple(out.shape) != (n, c, h, w) or out.dtype != dtype:
        raise RuntimeError(f"out= has shape {tuple(out.shape)} / {out.dtype}, producer writes {(n, c, h, w)} / {dtype}")
    return out


def geom(x):
    """(N, C, H, W, ld) of an NHWC-in-memory tensor; raises if the memory is not pixel-dense NHWC."""
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
    ok = (c == 1 or sc == 1) and (w == 1 or sw == ld) and (h == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld) \
        and ld >= c
    if not ok:
        raise RuntimeError(f"tensor is not NHWC-dense: shape {tuple(x.shape)} strides {x.stride()}")
    return n, c, h, w, ld


def is_nhwc(x):
    try:
        geom(x)
        return True
    except RuntimeError:
        return False


def to_nhwc(x, dtype):
    """Any NCHW-shaped tensor -> NHWC memory of `dtype` (tiled transpose kernels); no-op if already so."""
    if is_nhwc(x):
        if x.dtype == dtype:
            return x
        n, c, h, w, ld = geom(x)          # dtype change of NHWC memory: out through (n, c, m) and back
        tmp = torch.empty((n, c, h, w), dtype=dtype, device=x.device)
        lib.call("yolo_nhwc_to_ncm", _p(x), dt(x), ld, _p(tmp), dt(dtype), c * h * w, h * w, 0, n, c, h * w, _stream(x))
        x = tmp
    xc = x if x.is_contiguous() else x.contiguous()
    n, c, h, w = xc.shape
    out = new_nhwc(n, c, h, w, dtype, x.device)
    lib.call("yolo_ncm_to_nhwc", _p(xc), dt(xc), c * h * w, h * w, 0, _p(out), dt(dtype), c, n, c, h * w, _stream(xc))
    return out


def head_pack(x, preds, c_off, m_off):
    """preds[n, c_off:c_off+C, m_off:m_off+H*W] = x (NHWC) -- the (N, no, M) flattening of head.py:119."""
    n, c, h, w, ld = geom(x)
    cp, m = preds.shape[1], preds.shape[2]
    lib.call("yolo_nhwc_to_ncm", _p(x), dt(x), ld, preds.data_ptr() + c_off * m * preds.element_size(), dt(preds),
             cp * m, m, m_off, n, c, h * w, _stream(x))


def head_unpack(dpreds, c_off, c, m_off, h, w):
    n, cp, m = dpreds.shape
    out = new_nhwc(n, c, h, w, dpreds.dtype, dpreds.device)
    lib.call("yolo_ncm_to_nhwc", dpreds.data_ptr() + c_off * m * dpreds.element_size(), dt(dpreds), cp * m, m, m_off,
             _p(out), dt(out), c, n, c, h * w, _stream(dpreds))
    return out


def head_group(branches, preds, c_offs, m_offs, pack):
    """All branch tensors of the head <-> preds (N, no, M) in one launch: branch i occupies channels c_offs[i].. and anchors
    m_offs[i]..; pack=True writes preds, False fills the (preallocated) branches from it."""
    import ctypes
    k = len(branches)
    assert 1 <= k <= 8 and preds.is_contiguous() and all(b.dtype == preds.dtype for b in branches)
    n, cp, m = preds.shape
    gs = [geom(b) for b in branches]
    i32 = ctypes.c_int * k
    lib.call("yolo_head_group", int(pack), k, (ctypes.c_void_p * k)(*[b.data_ptr() for b in branches]), i32(*[g[4] for g in gs]),
             i32(*[g[1] for g in gs]), i32(*[g[2] * g[3] for g in gs]), i32(*c_offs), i32(*m_offs), _p(preds), cp, m, n,
             dt(preds), _stream(preds))


def copy_channels(src, dst, accumulate=False):
    n, c, h, w, lds = geom(src)
    n2, c2, h2, w2, ldd = geom(dst)
    assert (n, c, h, w) == (n2, c2, h2, w2) and src.dtype == dst.dtype
    lib.call("yolo_copy_channels", _p(src), lds, _p(dst), ldd, n * h * w, c, int(accumulate), dt(src), _stream(src))


def add_n(xs, out=None):
    """Sum of 2..4 same-shaped NHWC tensors (channel slices allowed) in one pass -> fresh NHWC tensor (or `out`)."""
    n, c, h, w, _ = geom(xs[0])
    out = _dest(out, n, c, h, w, xs[0].dtype, xs[0].device)
    g = [geom(x) for x in xs]
    assert all(gi[:4] == (n, c, h, w) for gi in g) and all(x.dtype == xs[0].dtype for x in xs) and 2 <= len(xs) <= 4
    ptr = [_p(x) for x in xs] + [0] * (4 - len(xs))
    ld = [gi[4] for gi in g] + [0] * (4 - len(xs))
    lib.call("yolo_add_n", ptr[0], ld[0], ptr[1], ld[1], ptr[2], ld[2], ptr[3], ld[3], len(xs), _p(out), geom(out)[4],
             n * h * w, c, dt(xs[0]), _stream(xs[0]))
    return out


def bucket_copy(plan, scale=1.0):
    """dst_i = src_i * scale (with dtype cast) for every tensor pair of a GradBuckets plan in ONE launch; `plan` carries
    the device job table built from the tensors' addresses (plus the tensors themselves, srcs / dsts)."""
    if plan["njobs"]:
        lib.call("yolo_multi_copy", _p(plan["dev"]), plan["njobs"], plan["nchunks"], float(scale), _stream(plan["dsts"][0]))


def zero_(t):
    lib.call("yolo_memset0", _p(t), t.numel() * t.element_size(), _stream(t))
    return t


def fill_(t, value):
    """Small constant vectors (fused-eval path only): plain tensor fill, not on the training path."""
    return t.fill_(value)


# ------------------------------------------------------------------------------------------------ conv
class WeightPackPlan:
    """Every dense conv weight of a model packed (forward + dgrad forms) by ONE launch per step.
    `run()` re-packs from the current parameter values; `lookup()` serves pack_weights() while the
    parameter has not been written since (tensor version check), so a stale plan can never be used."""

    def __init__(self, convs, dtype):
        """convs: list of (weight, k, stride) with plain (unsharded) OIHW parameters on one device."""
        dev = convs[0][0].device
        jb = lib.query("yolo_pack_job_bytes")
        sizes, njobs = [], 0
        for w, k, s in convs:
            o, i = w.shape[0], w.shape[1]
            sizes.append((o * lib.query("yolo_conv_kpad", o, i, k, s, 0, 0), lib.query("yolo_conv_dgrad_wbuf_elems", o, i, k, s)))
            njobs += 1 + lib.query("yolo_pack_job_count", s, 1)
        self.dtype, self.njobs = dtype, njobs
        self.total = sum(a + b for a, b in sizes)
        self.flat = torch.zeros(self.total, dtype=dtype, device=dev)     # pad columns stay zero: the pack kernel never writes them
        host = torch.zeros(njobs * jb, dtype=torch.uint8)
        self.entries, start, ji = {}, 0, 0
        esz = self.flat.element_size()
        for (w, k, s), (nf, nb) in zip(convs, sizes):
            o, i = w.shape[0], w.shape[1]
            for mode, n in ((0, nf), (1, nb)):
                view = self.flat[start:start + n]
                got = lib.query("yolo_pack_job_fill", host.data_ptr() + ji * jb, w.data_ptr(), dt(w), view.data_ptr(), esz,
                                o, i, k, s, mode, start)
                assert got == n, (got, n)
                self.entries[(w.data_ptr(), mode)] = [view, w, -1]
                ji += 1 if mode == 0 else lib.query("yolo_pack_job_count", s, 1)
                start += n
        self.nchunks = lib.query("yolo_pack_jobs_finalize", host.data_ptr(), njobs)
        self.jobs = host.to(dev)
        self.key = tuple(w.data_ptr() for w, _, _ in convs)

    def run(self):
        lib.call("yolo_pack_batched", _p(self.jobs), self.njobs, self.nchunks, dt(self.dtype), _stream(self.flat))
        for e in self.entries.values():
            e[2] = e[1]._version

    def lookup(self, w, mode, dtype):
        e = self.entries.get((w.data_ptr(), mode))
        if e is None or dtype != self.dtype or e[2] != w._version or w.shape != e[1].shape:
            return None
        return e[0]


ACTIVE_PACK_PLAN = None


def pack_weights(w, k, stride, mode, dtype):
    """OIHW parameter -> K-major packed matrix (mode 0 forward, 1 dgrad buffer) in the compute dtype."""
    if ACTIVE_PACK_PLAN is not None:
        hit = ACTIVE_PACK_PLAN.lookup(w, mode, dtype)
        if hit is not None:
            return hit
    o, i = w.shape[0], w.shape[1]
    w = w if w.is_contiguous() else w.contiguous()
    if mode == 0:
        elems = o * lib.query("yolo_conv_kpad", o, i, k, stride, 0, 0)
    else:
        elems = lib.query("yolo_conv_dgrad_wbuf_elems", o, i, k, stride)
    out = torch.empty(elems, dtype=dtype, device=w.device)
    lib.call("yolo_conv_pack_weights", _p(w), dt(w), o, i, k, stride, mode, _p(out), dt(dtype), _stream(w))
    return out


def conv_out_hw(h, w, k, stride):
    pad = k // 2
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def conv_fwd(x, wp, bias, cout, k, stride, stats_acc=None, out=None):
    """y = conv(x); with stats_acc (fp32 [8][2][cout], zeroed) also accumulates sum(y), sum(y^2) per channel."""
    n, cin, h, w, ldx = geom(x)
    oh, ow = conv_out_hw(h, w, k, stride)
    y = _dest(out, n, cout, oh, ow, x.dtype, x.device)
    lib.call("yolo_conv2d_fwd", _p(x), ldx, _p(wp), _p(bias), _p(y), geom(y)[4], _p(stats_acc), n, h, w, cin, oh, ow, cout, k,
             stride, dt(x), ALGO, _stream(x))
    return y


def conv_fwd_act(x, wp, bias, cout, k, stride, act, res=None, out=None):
    """act(conv(x) + bias) (+ res) in one launch (the fused inference block); None when the shape has no MFMA kernel."""
    n, cin, h, w, ldx = geom(x)
    oh, ow = conv_out_hw(h, w, k, stride)
    y = _dest(out, n, cout, oh, ow, x.dtype, x.device)
    ldr = geom(res)[4] if res is not None else 0
    rc = lib.query("yolo_conv2d_fwd_act", _p(x), ldx, _p(wp), _p(bias), _p(res), ldr, _p(y), geom(y)[4], n, h, w, cin, oh, ow, cout, k,
                   stride, int(act), dt(x), _stream(x))
    if rc == 1:
        return None
    lib.status(rc, "yolo_conv2d_fwd_act")
    return y


BN_REPL = 8


def bn_acc_new(c, device):
    """Zeroed statistics accumulator for one BN layer: fp32 [8][2][c]."""
    return zero_(torch.empty(BN_REPL * 2 * c, dtype=torch.float32, device=device))


def bn_stats_acc(y, acc):
    n, c, h, w, ld = geom(y)
    lib.call("yolo_bn_stats_acc", _p(y), ld, n * h * w, c, dt(y), _p(acc), _stream(y))


def bn_finalize_acc(acc, count, gamma, beta, running_mean, running_var, momentum, eps):
    """Accumulated (sum, sum of squares) -> (mean, invstd, scale, shift); updates the running buffers."""
    c = gamma.numel()
    coef = _f32(4 * c, gamma.device)
    mean, invstd, scale, shift = coef[:c], coef[c:2 * c], coef[2 * c:3 * c], coef[3 * c:]
    lib.call("yolo_bn_finalize_acc", _p(acc), count, c, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
             float(momentum), float(eps), _p(mean), _p(invstd), _p(scale), _p(shift), *_pb_dt(gamma, beta, running_mean, running_var),
             _stream(gamma))
    return mean, invstd, scale, shift


def bn_act_fwd_train(y, acc, gamma, beta, running_mean, running_var, momentum, eps, act, res=None, out=None):
    """act(BN(y)) (+res) with the batch statistics read from acc (finalize folded into the kernel's prologue);
    updates running stats; -> (out, mean, invstd, scale, shift)."""
    n, c, h, w, ld = geom(y)
    out = _dest(out, n, c, h, w, y.dtype, y.device)
    coef = _f32(4 * c, y.device)
    mean, invstd, scale, shift = coef[:c], coef[c:2 * c], coef[2 * c:3 * c], coef[3 * c:]
    ldr = geom(res)[4] if res is not None else 0
    lib.call("yolo_bn_act_fwd_train", _p(y), ld, _p(acc), n * h * w, _p(gamma), _p(beta), _p(running_mean),
             _p(running_var), float(momentum), float(eps), _p(mean), _p(invstd), _p(scale), _p(shift), _p(res), ldr, _p(out),
             geom(out)[4], n * h * w, c, int(act), dt(y), *_pb_dt(gamma, beta, running_mean, running_var), _stream(y))
    return out, mean, invstd, scale, shift


def bn_act_bwd_train(dout, y, scale, shift, mean, invstd, gamma, act, acc):
    """Backward of act(BN_batch(y)) in two launches: the reduction adds (sum dz, sum dz*y) into acc (fp32 [8][2][C],
    zeroed by the caller) with float atomics, the apply kernel folds the finalize into its prologue.
    -> (dy, dgamma, dbeta)."""
    n, c, h, w, ldy = geom(y)
    ldd = geom(dout)[4]
    npix = n * h * w
    st = _stream(y)
    lib.call("yolo_bn_bwd_reduce_acc", _p(dout), ldd, _p(y), ldy, _p(scale), _p(shift), npix, c, int(act), dt(y), _p(acc), st)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)        # in the parameter's own dtype
    dy = new_nhwc(n, c, h, w, y.dtype, y.device)
    lib.call("yolo_bn_act_bwd_apply_train", _p(dout), ldd, _p(y), ldy, _p(scale), _p(shift), _p(gamma), _p(mean), _p(invstd),
             _p(acc), npix, _p(dgamma), _p(dbeta), _p(dy), c, npix, c, int(act), dt(y), dt(gamma), st)
    return dy, dgamma, dbeta


def conv_dgrad(dy, wb, cin, h, w, k, stride, acc_into=None, acc2=None):
    """Data gradient; with `acc_into` (an NHWC tensor or channel-slice view of the input's shape) it is ADDED to that
    tensor in the kernel epilogue instead of being written to a new one (returns acc_into).  `acc2` (stride 1, with
    acc_into): a second tensor of that shape added in the same epilogue -- acc_into += dgrad + acc2."""
    n, cout, oh, ow, lddy = geom(dy)
    if acc2 is not None:
        if acc_into is None or stride != 1 or tuple(acc2.shape) != (n, cin, h, w) or acc2.dtype != dy.dtype \
                or tuple(acc_into.shape) != (n, cin, h, w) or acc_into.dtype != dy.dtype:
            raise RuntimeError("conv_dgrad: acc2 needs a stride-1 conv and acc_into / acc2 of the input gradient's shape / dtype")
        lib.call("yolo_conv2d_dgrad_acc2", _p(dy), lddy, _p(wb), _p(acc_into), geom(acc_into)[4], _p(acc2), geom(acc2)[4], n, h, w,
                 cin, oh, ow, cout, k, stride, dt(dy), ALGO, _stream(dy))
        return acc_into
    if acc_into is None:
        dx, accumulate = new_nhwc(n, cin, h, w, dy.dtype, dy.device), 0
    else:
        if tuple(acc_into.shape) != (n, cin, h, w) or acc_into.dtype != dy.dtype:
            raise RuntimeError("conv_dgrad: acc_into does not match the input gradient's shape / dtype")
        dx, accumulate = acc_into, 1
    lib.call("yolo_conv2d_dgrad", _p(dy), lddy, _p(wb), _p(dx), geom(dx)[4], n, h, w, cin, oh, ow, cout, k, stride,
             accumulate, dt(dy), ALGO, _stream(dy))
    return dx


def conv_wgrad(x, dy, k, stride, w_dtype, out=None):
    """OIHW weight gradient in w_dtype.  The kernels write per-slab partial matrices into a scratch buffer and a
    reduce pass sums them straight into the OIHW tensor (no memset, no atomics, no separate unpack)."""
    n, cin, h, w, ldx = geom(x)
    _, cout, oh, ow, ldy = geom(dy)
    args = (n, h, w, cin, oh, ow, cout, k, stride, dt(x), ALGO)
    ws = torch.empty(lib.query("yolo_conv2d_wgrad_ws_elems", _p(x), ldx, _p(dy), ldy, *args), dtype=torch.float32,
                     device=x.device)
    dw = torch.empty((cout, cin, k, k), dtype=w_dtype, device=x.device) if out is None else out
    assert dw.shape == (cout, cin, k, k) and dw.dtype == w_dtype and dw.is_contiguous()
    lib.call("yolo_conv2d_wgrad", _p(x), ldx, _p(dy), ldy, _p(ws), _p(dw), dt(w_dtype), *args, _stream(x))
    return dw


def stem_im2col(img, dtype, out=None):
    """(N,3,H,W) NCHW image (any float dtype) -> NHWC (N,32,OH,OW) column tensor: K = ci*9+kh*3+kw, 27..31 zero."""
    img = img if img.is_contiguous() else img.contiguous()
    n, c, h, w = img.shape
    assert c == 3
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    col = new_nhwc(n, 32, oh, ow, dtype, img.device) if out is None else out
    assert tuple(col.shape) == (n, 32, oh, ow) and col.dtype == dtype and geom(col)[4] == 32
    lib.call("yolo_stem_im2col", _p(img), dt(img), _p(col), dt(dtype), n, h, w, oh, ow, _stream(img))
    return col


def stem_conv_eligible(img, dtype, cout):
    return img.dtype == torch.float32 and dtype in _DT and \
        bool(lib.query("yolo_stem_conv_eligible", dt(img), dt(dtype), cout))


def stem_conv_fwd(img, wp, cout, dtype, stats_acc=None, bias=None, act=0):
    """(N,3,H,W) fp32 NCHW image -> NHWC (N,cout,OH,OW) of dtype: 3x3 stride-2 pad-1 conv with the [cout][32] matrix of
    stem_pack_weights, no column tensor; optionally accumulates the BatchNorm batch statistics like conv_fwd, or (fused
    inference, no statistics) applies act(y + bias) in the epilogue."""
    img = img if img.is_contiguous() else img.contiguous()
    n, c, h, w = img.shape
    assert c == 3
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = new_nhwc(n, cout, oh, ow, dtype, img.device)
    lib.call("yolo_stem_conv_fwd", _p(img), _p(wp), _p(y), cout, _p(stats_acc), n, h, w, oh, ow, cout, _p(bias), act, dt(dtype),
             _stream(img))
    return y


def stem_wgrad(img, dy, w_dtype, out=None):
    """Weight gradient (Cout,3,3,3) of the stem conv straight from the fp32 NCHW image and dy (NHWC, bf16/f16)."""
    n, _, h, w = img.shape
    _, cout, oh, ow, ldy = geom(dy)
    dw = torch.empty((cout, 3, 3, 3), dtype=w_dtype, device=img.device) if out is None else out
    assert dw.shape == (cout, 3, 3, 3) and dw.dtype == w_dtype and dw.is_contiguous() and img.is_contiguous()
    part = torch.empty(lib.query("yolo_stem_wgrad_slabs") * cout * 32, dtype=torch.float32, device=img.device)
    lib.call("yolo_stem_wgrad", _p(img), _p(dy), ldy, _p(part), _p(dw), dt(w_dtype), n, h, w, oh, ow, cout, dt(dy), _stream(img))
    return dw


def stem_pack_weights(w, dtype):
    """OIHW (Cout,3,3,3) -> forward-packed [Cout][32] matrix for conv_fwd(k=1) on the column tensor."""
    w = w if w.is_contiguous() else w.contiguous()
    out = torch.empty(w.shape[0] * 32, dtype=dtype, device=w.device)
    lib.call("yolo_stem_pack_weights", _p(w), dt(w), w.shape[0], _p(out), dt(dtype), _stream(w))
    return out


def stem_unpack_wgrad(dw32, w_dtype, out=None):
    """fp32 (Cout,32,1,1) gradient of the padded 1x1 weights -> (Cout,3,3,3) in the parameter dtype."""
    cout = dw32.shape[0]
    dw = torch.empty((cout, 3, 3, 3), dtype=w_dtype, device=dw32.device) if out is None else out
    assert dw.shape == (cout, 3, 3, 3) and dw.dtype == w_dtype and dw.is_contiguous()
    lib.call("yolo_stem_unpack_wgrad", _p(dw32), cout, _p(dw), dt(w_dtype), _stream(dw32))
    return dw


def dw_fwd(x, w9, stats_acc=None):
    """Depthwise 3x3; with `stats_acc` (a zeroed bn_acc_new buffer) the BatchNorm batch statistics of y are accumulated
    too -- by the conv kernel itself where it can, else by a statistics pass over y."""
    n, c, h, w, ldx = geom(x)
    y = new_nhwc(n, c, h, w, x.dtype, x.device)
    if stats_acc is not None:
        rc = lib.query("yolo_dwconv3x3_fwd_stats", _p(x), ldx, _p(w9), _p(y), c, _p(stats_acc), n, h, w, c, dt(x), _stream(x))
        if rc == 0:
            return y
        if rc != 1:
            lib.status(rc, "yolo_dwconv3x3_fwd_stats")
    lib.call("yolo_dwconv3x3_fwd", _p(x), ldx, _p(w9), _p(y), c, n, h, w, c, dt(x), _stream(x))
    if stats_acc is not None:
        bn_stats_acc(y, stats_acc)
    return y


def dw_fwd_act(x, w9, bias, act):
    """act(depthwise3x3(x) + bias) in one launch (a fused model's depthwise blocks); None when the strip kernel does not take
    the tensor (fp32, unaligned) -- the caller then runs dw_fwd + the element-wise pass."""
    n, c, h, w, ldx = geom(x)
    y = new_nhwc(n, c, h, w, x.dtype, x.device)
    rc = lib.query("yolo_dwconv3x3_fwd_act", _p(x), ldx, _p(w9), _p(bias), _p(y), c, n, h, w, c, act, dt(x), _stream(x))
    if rc == 1:
        return None
    lib.status(rc, "yolo_dwconv3x3_fwd_act")
    return y


def dw_dgrad(dy, w9, acc_into=None):
    """Depthwise data gradient; with `acc_into` it is ADDED to that tensor (gradient fan-in) instead of written to a new one."""
    n, c, h, w, ld = geom(dy)
    if acc_into is None:
        dx, accumulate = new_nhwc(n, c, h, w, dy.dtype, dy.device), 0
    else:
        if tuple(acc_into.shape) != (n, c, h, w) or acc_into.dtype != dy.dtype:
            raise RuntimeError("dw_dgrad: acc_into does not match the input gradient's shape / dtype")
        dx, accumulate = acc_into, 1
    lib.call("yolo_dwconv3x3_dgrad", _p(dy), ld, _p(w9), _p(dx), geom(dx)[4], n, h, w, c, accumulate, dt(dy), _stream(dy))
    return dx


def dw_wgrad(x, dy, out=None):
    n, c, h, w, ldx = geom(x)
    _, _, _, _, ldy = geom(dy)
    dw = torch.empty((c, 1, 3, 3), dtype=torch.float32, device=x.device) if out is None else out
    assert dw.shape == (c, 1, 3, 3) and dw.dtype == torch.float32 and dw.is_contiguous()
    part = torch.empty(lib.query("yolo_dw_wgrad_nslab", n, h) * c * 9, dtype=torch.float32, device=x.device)
    lib.call("yolo_dwconv3x3_wgrad", _p(x), ldx, _p(dy), ldy, _p(dw), _p(part), n, h, w, c, dt(x), _stream(x))
    return dw


# ------------------------------------------------------------------------------------------------ BN / act
def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


def _pb_dt(gamma, beta, running_mean, running_var):
    """(pdtype, bdtype) of a BatchNorm's parameters and buffers: the kernels read / write them in their own element type
    (fp32 normally; FSDP mixed precision hands over low-precision parameters and buffers: no cast launches)."""
    if beta.dtype != gamma.dtype or (running_mean is not None and running_var.dtype != running_mean.dtype):
        raise RuntimeError("BatchNorm: gamma / beta (and running_mean / running_var) must share a dtype")
    for t in (gamma, beta, running_mean, running_var):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("BatchNorm: parameters and buffers must be contiguous")
    return dt(gamma), dt(running_mean) if running_mean is not None else lib.F32


def bn_train_stats(y, gamma, beta, running_mean, running_var, momentum, eps):
    """Batch statistics of y; updates the running buffers in place; -> (mean, invstd, scale, shift)."""
    n, c, h, w, ld = geom(y)
    npix = n * h * w
    nblk = lib.query("yolo_reduce_nblk", npix, c)
    part = _f32(nblk * 2 * c, y.device)
    st = _stream(y)
    lib.call("yolo_bn_stats", _p(y), ld, npix, c, dt(y), _p(part), nblk, st)
    coef = _f32(4 * c, y.device)
    mean, invstd, scale, shift = coef[:c], coef[c:2 * c], coef[2 * c:3 * c], coef[3 * c:]
    lib.call("yolo_bn_finalize", _p(part), nblk, npix, c, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
             float(momentum), float(eps), _p(mean), _p(invstd), _p(scale), _p(shift), *_pb_dt(gamma, beta, running_mean, running_var), st)
    return mean, invstd, scale, shift


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    c = gamma.numel()
    coef = _f32(2 * c, gamma.device)
    lib.call("yolo_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), c,
             _p(coef[:c]), _p(coef[c:]), *_pb_dt(gamma, beta, running_mean, running_var), _stream(gamma))
    return coef[:c], coef[c:]


def bn_act_fwd(y, scale, shift, act, res=None, out=None):
    n, c, h, w, ld = geom(y)
    out = _dest(out, n, c, h, w, y.dtype, y.device)
    ldr = geom(res)[4] if res is not None else 0
    lib.call("yolo_bn_act_fwd", _p(y), ld, _p(scale), _p(shift), _p(res), ldr, _p(out), geom(out)[4], n * h * w, c, int(act),
             dt(y), _stream(y))
    return out


def bn_act_bwd(dout, y, scale, shift, mean, invstd, gamma, act):
    """Training-mode backward of act(BN(y)) -> (dy, dgamma, dbeta)."""
    n, c, h, w, ldy = geom(y)
    ldd = geom(dout)[4]
    npix = n * h * w
    nblk = lib.query("yolo_reduce_nblk", npix, c)
    part = _f32(nblk * 2 * c, y.device)
    st = _stream(y)
    lib.call("yolo_bn_act_bwd_reduce", _p(dout), ldd, _p(y), ldy, _p(scale), _p(shift), _p(mean), _p(invstd), npix, c,
             int(act), dt(y), _p(part), nblk, st)
    # dgamma / dbeta are returned to autograd: standalone tensors (a slice of a bigger buffer cannot be taken
    # over by AccumulateGrad and would be cloned with an extra copy kernel per parameter)
    dgamma, dbeta, coef = torch.empty_like(gamma), torch.empty_like(gamma), _f32(3 * c, y.device)
    lib.call("yolo_bn_bwd_finalize", _p(part), nblk, npix, c, _p(gamma), _p(mean), _p(invstd), _p(dgamma), _p(dbeta),
             _p(coef), dt(gamma), st)
    dy = new_nhwc(n, c, h, w, y.dtype, y.device)
    lib.call("yolo_bn_act_bwd_apply", _p(dout), ldd, _p(y), ldy, _p(scale), _p(shift), _p(mean), _p(invstd), _p(coef),
             _p(dy), c, npix, c, int(act), dt(y), st)
    return dy, dgamma, dbeta


def bn_act_bwd_eval(dout, y, scale, shift, act):
    """Backward through act(y*scale+shift) with frozen statistics -> dy."""
    n, c, h, w, ldy = geom(y)
    dy = new_nhwc(n, c, h, w, y.dtype, y.device)
    lib.call("yolo_bn_act_bwd_apply", _p(dout), geom(dout)[4], _p(y), ldy, _p(scale), _p(shift), 0, 0, 0, _p(dy), c,
             n * h * w, c, int(act), dt(y), _stream(y))
    return dy


def channel_sum(x, out=None):
    n, c, h, w, ld = geom(x)
    npix = n * h * w
    nblk = lib.query("yolo_reduce_nblk", npix, c)
    part = _f32(nblk * 2 * c, x.device)
    st = _stream(x)
    lib.call("yolo_bn_stats", _p(x), ld, npix, c, dt(x), _p(part), nblk, st)
    out = _f32(c, x.device) if out is None else out
    assert out.shape == (c,) and out.dtype == torch.float32
    lib.call("yolo_sum_finalize", _p(part), nblk, c, _p(out), st)
    return out


# ------------------------------------------------------------------------------------------------ pool / upsample
def maxpool5_fwd(x, out=None):
    n, c, h, w, ld = geom(x)
    out = _dest(out, n, c, h, w, x.dtype, x.device)
    idx = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device)
    lib.call("yolo_maxpool5_fwd", _p(x), ld, _p(out), geom(out)[4], _p(idx), n, h, w, c, dt(x), _stream(x))
    return out, idx


def _acc_dest(acc_into, n, c, h, w, like):
    if acc_into is None:
        return new_nhwc(n, c, h, w, like.dtype, like.device), 0
    if tuple(acc_into.shape) != (n, c, h, w) or acc_into.dtype != like.dtype:
        raise RuntimeError("acc_into does not match the gradient's shape / dtype")
    return acc_into, 1


def maxpool5_bwd(dout, idx, acc_into=None):
    n, c, h, w, ld = geom(dout)
    dx, accumulate = _acc_dest(acc_into, n, c, h, w, dout)
    lib.call("yolo_maxpool5_bwd", _p(dout), ld, _p(idx), _p(dx), geom(dx)[4], n, h, w, c, accumulate, dt(dout), _stream(dout))
    return dx


def upsample2x_fwd(x, out=None):
    n, c, h, w, ld = geom(x)
    out = _dest(out, n, c, 2 * h, 2 * w, x.dtype, x.device)
    lib.call("yolo_upsample2x_fwd", _p(x), ld, _p(out), geom(out)[4], n, h, w, c, dt(x), _stream(x))
    return out


def upsample2x_bwd(dout, acc_into=None):
    n, c, oh, ow, ld = geom(dout)
    dx, accumulate = _acc_dest(acc_into, n, c, oh // 2, ow // 2, dout)
    lib.call("yolo_upsample2x_bwd", _p(dout), ld, _p(dx), geom(dx)[4], n, oh // 2, ow // 2, c, accumulate, dt(dout), _stream(dout))
    return dx


# ------------------------------------------------------------------------------------------------ attention
def attn_fwd(qkv, heads, dk, dh, scale):
    """-> (o, vp, stash); stash = what the backward needs (fp32: row log-sum-exp, 16-bit: probabilities)."""
    n, cq, h, w, ld = geom(qkv)
    t = h * w
    o = new_nhwc(n, heads * dh, h, w, qkv.dtype, qkv.device)
    vp = new_nhwc(n, heads * dh, h, w, qkv.dtype, qkv.device)
    stash = torch.empty(lib.query("yolo_attn_stash_bytes_for", n, t, heads, dk, dh, dt(qkv)), dtype=torch.uint8, device=qkv.device)
    ws = torch.empty(lib.query("yolo_attn_workspace_bytes_for", n, t, heads, dk, dh, dt(qkv)), dtype=torch.uint8, device=qkv.device)
    lib.call("yolo_attn_fwd", _p(qkv), ld, _p(o), heads * dh, _p(vp), heads * dh, _p(stash), _p(ws), n, t, heads, dk, dh,
             float(scale), dt(qkv), _stream(qkv))
    return o, vp, stash


def attn_fwd_nograd(qkv, heads, dk, dh, scale):
    """-> (o, vp) with nothing kept for a backward, or None when the fused kernels do not take the problem (fp32, other
    head shapes): any sequence length, no score / probability matrix in memory."""
    n, cq, h, w, ld = geom(qkv)
    o = new_nhwc(n, heads * dh, h, w, qkv.dtype, qkv.device)
    vp = new_nhwc(n, heads * dh, h, w, qkv.dtype, qkv.device)
    rc = lib.query("yolo_attn_fwd_nograd", _p(qkv), ld, _p(o), heads * dh, _p(vp), heads * dh, n, h * w, heads, dk, dh, float(scale),
                   dt(qkv), _stream(qkv))
    if rc == 1:
        return None
    lib.status(rc, "yolo_attn_fwd_nograd")
    return o, vp


def attn_bwd(qkv, o, d_o, d_vp, stash, heads, dk, dh, scale):
    n, cq, h, w, ld = geom(qkv)
    t = h * w
    dqkv = new_nhwc(n, cq, h, w, qkv.dtype, qkv.device)
    ws = torch.empty(lib.query("yolo_attn_workspace_bytes_for", n, t, heads, dk, dh, dt(qkv)), dtype=torch.uint8, device=qkv.device)
    lib.call("yolo_attn_bwd", _p(qkv), ld, _p(o), geom(o)[4], _p(d_o), geom(d_o)[4], _p(d_vp),
             geom(d_vp)[4] if d_vp is not None else 0, _p(stash), _p(ws), _p(dqkv), cq, n, t, heads, dk, dh,
             float(scale), dt(qkv), _stream(qkv))
    return dqkv


# ------------------------------------------------------------------------------------------------ loss / post
def loss_fwd_bwd(preds, anchors, strides, gt, gt_off, gt_img, n_gt, nc, lambda_dfl, lambda_cls, want_grad, grad_scale=None):
    """-> (out[3] fp32 = total, mean_dfl, mean_cls ; dpreds or None).  `grad_scale`: optional fp32 device scalar the
    gradient is multiplied by in fp32 before it is rounded to the prediction dtype (fp16 loss scaling)."""
    assert grad_scale is None or (grad_scale.dtype == torch.float32 and grad_scale.is_cuda)
    n, cp, a = preds.shape
    assert preds.is_contiguous() and cp == 64 + nc
    anchors = anchors.to(preds.dtype).contiguous()
    strides = strides.to(preds.dtype).contiguous()
    ws = torch.empty(lib.query("yolo_loss_workspace_bytes", n, a, n_gt), dtype=torch.uint8, device=preds.device)
    out = _f32(3, preds.device)
    dpreds = torch.empty_like(preds) if want_grad else None
    lib.call("yolo_loss_dfl_qfl", _p(preds), _p(anchors), _p(strides), dt(preds), n, nc, a, _p(gt), _p(gt_off),
             _p(gt_img), n_gt, float(lambda_dfl), float(lambda_cls), _p(dpreds), _p(out), _p(ws), _p(grad_scale), _stream(preds))
    return out, dpreds, ws


def _f32c(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("the stand-alone loss helpers take contiguous fp32 tensors")
    return t


def bbox_iou(b1, b2, g=None):
    """forward (g None) -> iou [M]; backward -> (db1, db2) = g * d iou / d box  (losses.py:9-40 incl. the b1_y2 slip)"""
    m = b1.shape[0]
    _f32c(b1), _f32c(b2)
    if g is None:
        out = _f32(m, b1.device)
        lib.call("yolo_bbox_iou", _p(b1), _p(b2), m, _p(out), 0, 0, 0, _stream(b1))
        return out
    d1, d2 = torch.empty_like(b1), torch.empty_like(b2)
    lib.call("yolo_bbox_iou", _p(b1), _p(b2), m, 0, _p(_f32c(g)), _p(d1), _p(d2), _stream(b1))
    return d1, d2


def qfl(pred, target, beta, g=None):
    """quality_focal_loss (losses.py:46-57): forward -> 0-d loss; backward (g: 0-d device tensor) -> (dpred, dtarget)"""
    m, c = pred.shape
    _f32c(pred), _f32c(target)
    if g is None:
        out = _f32(1, pred.device)
        lib.call("yolo_quality_focal_loss", _p(pred), _p(target), m, c, float(beta), _p(out), 0, 0, 0, _stream(pred))
        return out.reshape(())
    dp, dtg = torch.empty_like(pred), torch.empty_like(target)
    lib.call("yolo_quality_focal_loss", _p(pred), _p(target), m, c, float(beta), 0, _p(_f32c(g)), _p(dp), _p(dtg), _stream(pred))
    return dp, dtg


def dfl_loss(pred_dist, target_val, g=None):
    """distribution_focal_loss (losses.py:63-78): forward -> 0-d loss; backward -> (dpred, dtarget)"""
    m, c = pred_dist.shape
    _f32c(pred_dist), _f32c(target_val)
    if g is None:
        out = _f32(1, pred_dist.device)
        lib.call("yolo_distribution_focal_loss", _p(pred_dist), _p(target_val), m, c, _p(out), 0, 0, 0, _stream(pred_dist))
        return out.reshape(())
    dp, dtg = torch.empty_like(pred_dist), torch.empty_like(target_val)
    lib.call("yolo_distribution_focal_loss", _p(pred_dist), _p(target_val), m, c, 0, _p(_f32c(g)), _p(dp), _p(dtg), _stream(pred_dist))
    return dp, dtg


def scale_inplace(x, scale_dev):
    lib.call("yolo_scale_inplace", _p(x), x.numel(), dt(x), _p(scale_dev), _stream(x))
    return x


def image_prep(src_u8, recs, size, jitter, dtype, mean, std):
    """Batch of decoded uint8 HWC images (flat device buffer `src_u8`; recs = [(offset, H, W, flip, order[4], factors[4])])
    -> normalised (N, 3, size, size) tensor of `dtype`: flip + antialiased bilinear resize + colour jitter + normalise
    (src/data/transforms.py:4-24) in 2 + 2 x (jitter slots) launches for the whole batch."""
    n = len(recs)
    rb = lib.query("yolo_prep_image_bytes")
    host = torch.zeros(n * rb, dtype=torch.uint8).pin_memory()
    for i, (off, h, w, flip, order, fac) in enumerate(recs):
        o = list(order) + [-1] * (4 - len(order))
        lib.call("yolo_prep_image_fill", host.data_ptr(), i, int(off), int(h), int(w), int(bool(flip)), *[int(v) for v in o],
                 *[float(v) for v in fac])
    table = host.to(src_u8.device, non_blocking=True)
    stage = torch.empty((n, 3, size, size), dtype=torch.uint8, device=src_u8.device)
    means = torch.empty(4 * n, dtype=torch.float32, device=src_u8.device)
    out = torch.empty((n, 3, size, size), dtype=dtype, device=src_u8.device)
    lib.call("yolo_image_prep", _p(src_u8), _p(table), n, size, int(bool(jitter)), _p(stage), _p(means), _p(out), dt(dtype),
             *[float(v) for v in mean], *[float(v) for v in std], _stream(src_u8))
    return out


def head_decode(preds, anchors, strides, nc):
    n, cp, a = preds.shape
    preds = preds.contiguous()
    y = torch.empty((n, 4 + nc, a), dtype=preds.dtype, device=preds.device)
    # keep the converted copies alive in locals: a temporary freed before the launch may be handed to the
    # next allocation and overwritten
    anc = anchors.to(preds.dtype).contiguous()
    strd = strides.to(preds.dtype).contiguous()
    lib.call("yolo_head_decode", _p(preds), _p(anc), _p(strd), _p(y), n, nc, a, dt(preds), _stream(preds))
    return y


def dfl_expect(x):
    b, c, a = x.shape
    x = x.contiguous()
    y = torch.empty((b, 4, a), dtype=x.dtype, device=x.device)
    lib.call("yolo_dfl_expect", _p(x), _p(y), b, a, dt(x), _stream(x))
    return y


def nms(y, nc, conf_thres, iou_thres, classes, agnostic, multi_label, max_det):
    """-> (rows fp32 [bs][max_det][6], counts int32 [bs], status int32 [1])."""
    import ctypes
    bs, _, m = y.shape
    y = y.contiguous()
    ws = torch.empty(lib.query("yolo_nms_workspace_bytes", bs, m, nc, int(multi_label)), dtype=torch.uint8, device=y.device)
    rows = zero_(torch.empty((bs, max_det, 6), dtype=torch.float32, device=y.device))
    counts = zero_(torch.empty(bs, dtype=torch.int32, device=y.device))
    status = zero_(torch.empty(1, dtype=torch.int32, device=y.device))
    cl = list(classes) if classes is not None else []
    arr = (ctypes.c_int * max(1, len(cl)))(*cl)
    lib.call("yolo_nms", _p(y), dt(y), bs, nc, m, float(conf_thres), float(iou_thres), ctypes.cast(arr, ctypes.c_void_p),
             len(cl), int(agnostic), int(multi_label), int(max_det), _p(rows), _p(counts), _p(status), _p(ws), _stream(y))
    return rows, counts, status


def val_select(y, nc, conf_threshold, top_k):
    """decode_predictions' per-image selection on the decoded head output y (N, 4+nc, M):
    -> (rows fp32 [N][top_k][6] = cx, cy, w, h, cls, score (zero padded), counts int32 [N])."""
    n, _, m = y.shape
    y = y.contiguous()
    ws = torch.empty(lib.query("yolo_val_workspace_bytes", n, m, top_k), dtype=torch.uint8, device=y.device)
    rows = torch.empty((n, top_k, 6), dtype=torch.float32, device=y.device)
    count = torch.empty(n, dtype=torch.int32, device=y.device)
    lib.call("yolo_val_select", _p(y), dt(y), n, nc, m, float(conf_threshold), top_k, _p(rows), _p(count), _p(ws), _stream(y))
    return rows, count


def val_match(rows, count, gt, gt_off, iou_threshold, nc, skip_empty_gt, counters, status):
    """DetectionMetrics.update for a batch: rows/count from val_select (or any fp32 [N][K][>=5 of 6] rows), gt fp32
    [total][5], gt_off int32 [N+1]; accumulates into the int64 counters [5 + 4 nc]."""
    n, k, six = rows.shape
    assert six == 6 and rows.dtype == torch.float32 and rows.is_contiguous()
    assert gt.dtype == torch.float32 and gt.is_contiguous() and gt_off.dtype == torch.int32 and count.dtype == torch.int32
    assert counters.dtype == torch.int64 and counters.numel() == 5 + 4 * nc
    lib.call("yolo_val_match", _p(rows), _p(count), n, k, _p(gt), _p(gt_off), float(iou_threshold), nc, int(skip_empty_gt),
             _p(counters), _p(status), _stream(rows))
