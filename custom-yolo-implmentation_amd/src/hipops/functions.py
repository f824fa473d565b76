"""torch.autograd Functions over the HIP leaf ops (ops.py).  Each Function is the fused unit the
reference spells as several ATen ops; the forward saves exactly what its hand-written backward needs.
All leaf calls go through the `ops` module namespace (tests swap leaves for a torch emulation to check
this wiring on a machine without a GPU; the product never does)."""
import os

import torch

from . import ops
from .lib import ACT_IDENTITY, ACT_SILU  # noqa: F401

_LOWP = (torch.bfloat16, torch.float16)


def compute_dtype(x, w):
    """Element type the block computes in: the autocast dtype when autocast is on (DDP mode of the
    reference, src/training/train_model.py:240-243), else the parameters' low-precision dtype (FSDP
    mixed precision casts them, src/training/utils_train.py:84-89,146-149), else the input's."""
    dev = x.device.type
    if torch.is_autocast_enabled(dev):
        return torch.get_autocast_dtype(dev)
    if w.dtype in _LOWP:
        return w.dtype
    return x.dtype if x.dtype in (torch.float32,) + _LOWP else torch.float32


def _as_nhwc(t, dtype):
    return t if (t.dtype == dtype and ops.is_nhwc(t)) else ops.to_nhwc(t, dtype)


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


DETERMINISTIC = os.environ.get("YOLO_DETERMINISTIC", "") == "1"


def deterministic_stats():
    """Process-wide deterministic mode: `torch.use_deterministic_algorithms(True)` (or YOLO_DETERMINISTIC=1, or
    `functions.DETERMINISTIC = True`).  The default training path sums the BatchNorm batch statistics with 8-way
    replicated float atomics (conv / depthwise / stem epilogues, k_channel_acc), whose order differs from run to run;
    in this mode every statistic goes through the fixed-order two-level reduction, so a step is bit-reproducible (the
    reference's CPU path is).  Cost: one more read of every conv output in forward, two finalize launches per layer."""
    return DETERMINISTIC or torch.are_deterministic_algorithms_enabled()


_SIDE = {}
FOLD_BN_FINALIZE = True     # BN scale/shift are derived in the prologue of the activation kernel (no finalize launch)
OVERLAP_WGRAD = True        # run a conv's weight gradient on a side stream, concurrently with its data gradient
FWD_STREAM = None           # see ConvBnAct.forward
HEAD_TWO_STREAMS = os.environ.get("YOLO_HEAD_STREAMS", "1") == "1"       # Head.forward; 0 for A/B runs
LAZY_WGRAD_JOIN = False     # set by a caller that owns the whole backward (TrainStepRunner) and joins at its end
# convs per cross-stream sync point in lazy mode.  Measured on preset s (img/s): per-layer fork/join 2238, 1: 2275,
# 2: 2320, 4: 2339, 8: 2354-2368, 12: 2330, 16: 2322, 32: 2312, all at the end: 2217 (dy has left the caches by then);
# grouping by queued bytes instead of by count was worse
WGRAD_GROUP = int(os.environ.get("YOLO_WGRAD_GROUP", "8"))
_INFLIGHT = {}              # device -> [(keepalive, fn)] of the work still running on the side stream
_QUEUED = {}                # device -> [(keepalive, fn)] not yet issued


def side_stream(dev):
    """The one auxiliary HIP stream per device (graph capture of more forked streams is not reliable on this stack)."""
    side = _SIDE.get(dev)
    if side is None:
        side = _SIDE[dev] = torch.cuda.Stream(dev)
    return side


_STEP = {}


def step_stream(dev):
    """The one non-default stream per device on which captured training steps are warmed up and captured -- and on which
    `prepare_ddp_model` constructs DistributedDataParallel: the wrapper keeps every parameter's AccumulateGrad node
    alive, and autograd runs such a node on the stream it was CREATED on.  Created on the default stream (a plain
    `DDP(model)`), they drag the default stream into the capture and hipStreamEndCapture dies (tools/capture_probe.py);
    created on this stream they run where the rest of the captured backward runs."""
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _STEP.get(dev)
    if st is None:
        st = _STEP[dev] = torch.cuda.Stream(dev)
    return st


JOIN_EACH_GROUP = os.environ.get("YOLO_WGRAD_JOIN", "1") == "1"


def _issue_wgrads(dev, cur, side):
    """One sync point: join what runs on the side stream, fork, issue every queued piece of work there.
    The join releases the previous group's inputs (x, dy) and costs the main chain 35-125 us of idle time at most sync
    points of a step (tools/queue_gaps.py: 0.6-0.9 ms per step: the side stream has not drained yet).  Dropping it
    (YOLO_WGRAD_JOIN=0: inputs referenced until join_wgrad_stream, one join at the end) was measured SLOWER in the replayed
    graph, 12.32 vs 11.70 ms per step on the same box, at every group size: the side stream then lags without bound and
    shares the chip with the main chain all the way through the large-map layers at the end of backward."""
    jobs = _QUEUED.pop(dev, [])
    if JOIN_EACH_GROUP:
        if _INFLIGHT.get(dev):
            cur.wait_stream(side)
        _INFLIGHT[dev] = jobs               # the previous group's inputs may be reused from here on
    else:
        _INFLIGHT.setdefault(dev, []).extend(jobs)
    if jobs:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _, fn in jobs:
                fn()


def defer_to_side(dev, keepalive, fn):
    """In a LAZY_WGRAD_JOIN backward: queue `fn` (parameter-gradient work nothing downstream in backward reads; it
    must write into tensors allocated by the caller) for the side stream and return True; `keepalive` holds its
    inputs until the join after it ran.  Otherwise return False (the caller runs it inline)."""
    if not (LAZY_WGRAD_JOIN and OVERLAP_WGRAD and dev.type == "cuda"):
        return False
    side = side_stream(dev)
    q = _QUEUED.setdefault(dev, [])
    q.append((keepalive, fn))
    if len(q) >= WGRAD_GROUP:
        _issue_wgrads(dev, torch.cuda.current_stream(dev), side)
    return True


def _wgrad_overlapped(x, dy, k, stride, w_dtype, dgrad_fn):
    """(dx, dw) of a dense conv.  The two gradients are independent: the weight gradient is issued on a side HIP
    stream (fork after dy is ready, join before returning), so on the small maps -- kernels of 100-400 workgroups on
    a 256-CU chip -- the two run side by side; inside a captured step this becomes a fork/join in the hipGraph.
    A cross-stream sync point costs ~10 us of idle GPU in the replayed graph (tools/trace_gaps.py).  With
    LAZY_WGRAD_JOIN the join of a layer is taken at the fork of a LATER conv's backward (the side stream has long
    finished by then) and WGRAD_GROUP convs share one sync point; x and dy of queued / running weight gradients are
    kept alive until that join, and dw is valid only after join_wgrad_stream()."""
    if not (OVERLAP_WGRAD and x.is_cuda):
        return dgrad_fn(), ops.conv_wgrad(x, dy, k, stride, w_dtype)
    dev = x.device
    cur, side = torch.cuda.current_stream(dev), side_stream(dev)
    if LAZY_WGRAD_JOIN:
        dw = torch.empty((dy.shape[1], x.shape[1], k, k), dtype=w_dtype, device=dev)
        fn = lambda: ops.conv_wgrad(x, dy, k, stride, dw.dtype, out=dw)
        if not defer_to_side(dev, (x, dy), fn):
            fn()
        # a second tensor object on the same storage: autograd's AccumulateGrad clones a gradient it cannot steal
        # (use count > 1), and that clone would read dw before the side stream has written it
        return dgrad_fn(), dw.detach()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        dw = ops.conv_wgrad(x, dy, k, stride, w_dtype)
    dx = dgrad_fn()
    cur.wait_stream(side)
    dw.record_stream(cur)           # allocated on the side stream, consumed (optimizer / reducer) on this one
    return dx, dw


def join_wgrad_stream(device):
    """End of a LAZY_WGRAD_JOIN backward: issue what is still queued and wait for the side stream."""
    device = torch.device(device)
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    side = _SIDE.get(device)
    if side is None:
        return
    cur = torch.cuda.current_stream(device)
    _issue_wgrads(device, cur, side)
    if _INFLIGHT.get(device):
        cur.wait_stream(side)
    _INFLIGHT[device] = []


class BnArena:
    """Per-forward pool of zeroed BatchNorm accumulators: ONE memset covers every layer of a model pass.
    A fresh pool per Model.forward keeps un-backpropagated passes independent (the autograd graph keeps
    its pool alive); Conv blocks outside a Model allocate their own pair."""
    current = None

    def __init__(self, device, elems):
        self.buf = ops.zero_(torch.empty(max(elems, 1), dtype=torch.float32, device=device))
        self.off = 0

    def take(self, c):
        n = ops.BN_REPL * 2 * c
        if self.off + n > self.buf.numel():
            return None
        acc = self.buf[self.off:self.off + n]
        self.off += n
        return acc

    @staticmethod
    def elems_for(channels):
        return 2 * sum(ops.BN_REPL * 2 * c for c in channels)      # forward statistics + backward sums per layer


class ResLink:
    """Shared by the two convs of x + conv2(conv1(x)): conv2's backward leaves the residual gradient here and conv1's
    data gradient is ADDED to it in its kernel epilogue, instead of autograd summing two tensors with one more pass
    over them.  Safe only if nothing else consumes x between the two: Residual wraps its input in `Alias`, whose only
    consumers are those two convs.

    `fan=True` (see `fan2`) makes the protocol symmetric for ANY two consumers of an alias: whichever backward runs
    first leaves its gradient in `dres` (and returns it to autograd), the second ADDS its own into that tensor in its
    kernel epilogue and returns None -- autograd never sums, no extra pass over the gradient."""
    __slots__ = ("dres", "fan", "chunk")

    def __init__(self, fan=False):
        self.dres = None
        self.fan = fan
        self.chunk = None           # ChunkLink of an enclosing C3K2 whose chunk half this alias is (see ChunkLink)

    def usable(self, shape, dtype):
        """The gradient left by the first consumer, if the second can accumulate into it."""
        d = self.dres
        if d is None or tuple(d.shape) != tuple(shape) or d.dtype != dtype or not ops.is_nhwc(d):
            return None
        return d


CHUNK_LINK = os.environ.get("YOLO_CHUNK_LINK", "1") == "1"     # 0: A/B runs


class ChunkLink:
    """C3K2's second chunk half feeds the concat (gradient = a slice of the concat's gradient, `dst`, known as soon as
    CatInto.backward ran) AND the first chained block.  The block's entry convs add their data gradients straight into
    `dst` -- a Residual's conv1 together with the skip gradient (`ops.conv_dgrad(acc_into=dst, acc2=skip gradient)`), a
    C3K's two 1x1 convs one after the other -- and set `done`; Chunk2.backward then has nothing left to accumulate
    (one pass over the gradient less per C3K2)."""
    __slots__ = ("dst", "done")

    def __init__(self):
        self.dst = None
        self.done = False


class Alias(torch.autograd.Function):
    """The same tensor under a new autograd node: consumers of the alias are the alias' only consumers.  With a link,
    the node (which runs after both consumers' backward) clears it: a second backward over a retained graph starts clean."""

    @staticmethod
    def forward(ctx, x, link=None):
        ctx.link = link
        return _fresh(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.link is not None:
            ctx.link.dres = None
        return g, None


class Stash(torch.autograd.Function):
    """Handle on an alias for a consumer that has no accumulating epilogue of its own (a concat slice, another fan-out):
    its gradient goes into the link for the other consumer to add to; if the other consumer came first, it is added to
    that one here (one HIP pass)."""

    @staticmethod
    def forward(ctx, x, link):
        ctx.link = link
        return _fresh(x)

    @staticmethod
    def backward(ctx, g):
        link = ctx.link
        g = _as_nhwc(g, g.dtype)
        first = link.usable(g.shape, g.dtype)
        if first is None:
            link.dres = g
            return g, None
        ops.copy_channels(g, first, accumulate=True)
        return None, None


def fan2(x):
    """(alias of x, link) for exactly TWO consumers of the alias that each take the link: `conv(alias, res_link=link)`,
    `Upsample2x / MaxPool5 .apply(alias, out, link)`, or `Stash.apply(alias, link)` for anything else.  Without a gradient
    to propagate: (x, None)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, None
    link = ResLink(fan=True)
    return Alias.apply(x, link), link


def stash(x, link):
    return x if link is None else Stash.apply(x, link)


class Fanout(torch.autograd.Function):
    """x handed to k consumers as k aliases: autograd delivers the k gradients TOGETHER and they are summed by one HIP
    pass (ops.add_n: fp32 accumulate, one rounding) instead of k-1 ATen adds of three memory passes each."""

    @staticmethod
    def forward(ctx, x, k):
        return tuple(_fresh(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        T = gs[0].dtype
        gs = [_as_nhwc(g, T) for g in gs]
        acc = ops.add_n(gs[:4])
        for i in range(4, len(gs), 3):
            acc = ops.add_n([acc] + gs[i:i + 3])
        return acc, None


def fanout(x, k):
    """k handles on x for k consumers (see Fanout); plain references when no gradient will flow."""
    if k < 2 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * k
    return Fanout.apply(x, k)


class ConvBnAct(torch.autograd.Function):
    """act(BN(conv(x))) (+ residual).  Reference: Conv.forward, src/model/model_blocks.py:31-34; the
    residual adds of Residual/PSABlock (:62, :223-224) ride in the same epilogue.
    groups is 1 or C (depthwise 3x3).  Training: the conv epilogue accumulates the batch statistics (no
    separate pass over y); the backward uses the deterministic two-level reduction."""

    @staticmethod
    def forward(ctx, *args):
        # FWD_STREAM: launch this node's kernels on another stream while autograd still sees it on the current one
        # (its backward then runs on the current stream; see Head.forward)
        if FWD_STREAM is not None:
            with torch.cuda.stream(FWD_STREAM):
                return ConvBnAct._forward(ctx, *args)
        return ConvBnAct._forward(ctx, *args)

    @staticmethod
    def _forward(ctx, x, weight, gamma, beta, res, bufs, k, stride, depthwise, act, training, momentum, eps, out=None,
                 res_link=None):
        ctx.res_link = res_link
        T = compute_dtype(x, weight)
        cout = weight.shape[0]
        acc_f = None
        # deterministic mode: no float atomics anywhere -- the statistics come from the fixed-order two-level reduction
        # (k_channel_reduce + finalize) instead of the conv epilogue, forward and backward
        deterministic = training and deterministic_stats()
        if training and not deterministic:
            acc_f = BnArena.current.take(cout) if BnArena.current is not None else None
            if acc_f is None:
                acc_f = ops.bn_acc_new(cout, x.device)
        # stem: a 3-channel image feeding a 3x3/2 conv is unfolded once (from NCHW directly) and then
        # runs as a 1x1 conv over K = 32 columns; x (saved for wgrad) becomes that column tensor
        stem = (not depthwise and weight.shape[1] == 3 and k == 3 and stride == 2 and not ctx.needs_input_grad[0])
        # On the device the conv reads the fp32 NCHW image directly (no column tensor) and so does its weight gradient
        # (ops.stem_wgrad); the unfold + 1x1 route remains for dtypes / channel counts the fused kernels do not take.
        stem_fused = stem and x.is_cuda and ops.stem_conv_eligible(x, T, cout)
        if stem_fused:
            x = x if x.is_contiguous() else x.contiguous()
            y = ops.stem_conv_fwd(x, ops.stem_pack_weights(weight, T), cout, T, acc_f)
        elif stem:
            x = ops.stem_im2col(x, T)
            y = ops.conv_fwd(x, ops.stem_pack_weights(weight, T), None, cout, 1, 1, acc_f)
        elif depthwise:
            x = _as_nhwc(x, T)
            y = ops.dw_fwd(x, _f32(weight).reshape(cout, 9), acc_f)
        else:
            x = _as_nhwc(x, T)
            y = ops.conv_fwd(x, ops.pack_weights(weight, k, stride, 0, T), None, cout, k, stride, acc_f)
        # gamma / beta / running statistics go to the kernels in their own dtype (fp32; the low-precision parameter and
        # buffer dtype under FSDP mixed precision, src/training/utils_train.py:84-89,146-153): no cast launches
        rm, rv = bufs
        if res is not None:
            res = _as_nhwc(res, T)
        acc_b = None
        if training:
            if FOLD_BN_FINALIZE and BnArena.current is not None and not deterministic:
                acc_b = BnArena.current.take(cout)          # zeroed with the forward statistics; used by the backward
            if deterministic:
                mean, invstd, scale, shift = ops.bn_train_stats(y, gamma, beta, rm, rv, momentum, eps)
                out = ops.bn_act_fwd(y, scale, shift, act, res, out)
            elif FOLD_BN_FINALIZE:
                out, mean, invstd, scale, shift = ops.bn_act_fwd_train(y, acc_f, gamma, beta, rm, rv, momentum, eps, act, res, out)
            else:
                count = y.shape[0] * y.shape[2] * y.shape[3]
                mean, invstd, scale, shift = ops.bn_finalize_acc(acc_f, count, gamma, beta, rm, rv, momentum, eps)
                out = ops.bn_act_fwd(y, scale, shift, act, res, out)
        else:
            mean = invstd = None
            scale, shift = ops.bn_eval_coeffs(gamma, beta, rm, rv, eps)
            out = ops.bn_act_fwd(y, scale, shift, act, res, out)
        saved = (scale, shift, mean, invstd, gamma)
        ctx.acc_b = acc_b       # a slice of the arena other layers write to: kept off save_for_backward's version check
        ctx.cfg = (k, stride, depthwise, act, training, tuple(x.shape), res is not None, gamma.dtype, stem)
        ctx.stem_fused = stem_fused
        ctx.save_for_backward(x, weight, y, *saved)
        return _fresh(out)

    @staticmethod
    def backward(ctx, dout):
        k, stride, depthwise, act, training, xshape, has_res, gdtype, stem = ctx.cfg
        x, weight, y = ctx.saved_tensors[:3]
        T = y.dtype
        dout = _as_nhwc(dout, T)
        scale, shift, mean, invstd, gamma = ctx.saved_tensors[3:]
        acc_b = ctx.acc_b
        if training and acc_b is not None:
            dy, dgamma, dbeta = ops.bn_act_bwd_train(dout, y, scale, shift, mean, invstd, gamma, act, acc_b)
        elif training:
            dy, dgamma, dbeta = ops.bn_act_bwd(dout, y, scale, shift, mean, invstd, gamma, act)
        else:
            dy, dgamma, dbeta = ops.bn_act_bwd_eval(dout, y, scale, shift, act), None, None
        dx = dw = None
        n, cin, h, w = xshape
        if stem:
            if ctx.needs_input_grad[1]:
                def stem_dw(out=None):          # x is the fp32 image (fused) or the unfolded column tensor
                    if ctx.stem_fused:
                        return ops.stem_wgrad(x, dy, weight.dtype, out)
                    return ops.stem_unpack_wgrad(ops.conv_wgrad(x, dy, 1, 1, torch.float32), weight.dtype, out)
                if x.is_cuda and LAZY_WGRAD_JOIN:
                    dwb = torch.empty(weight.shape, dtype=weight.dtype, device=x.device)
                    if not defer_to_side(x.device, (x, dy), lambda: stem_dw(dwb)):
                        stem_dw(dwb)
                    dw = dwb.detach()
                else:
                    dw = stem_dw()
        elif depthwise:
            if ctx.needs_input_grad[0]:
                link = ctx.res_link if (ctx.res_link is not None and ctx.res_link.fan and not has_res) else None
                into = link.usable((n, cin, h, w), T) if link is not None else None
                dx = ops.dw_dgrad(dy, _f32(weight).reshape(weight.shape[0], 9), acc_into=into)
                if into is not None:
                    dx = None                                   # already inside the gradient the first consumer returned
                elif link is not None:
                    link.dres = dx
            if ctx.needs_input_grad[1]:
                if weight.dtype == torch.float32 and x.is_cuda and LAZY_WGRAD_JOIN:
                    dwb = torch.empty((weight.shape[0], 1, 3, 3), dtype=torch.float32, device=x.device)
                    fn = lambda: ops.dw_wgrad(x, dy, out=dwb)
                    if not defer_to_side(x.device, (x, dy), fn):
                        fn()
                    dw = dwb.detach()
                else:
                    dw = ops.dw_wgrad(x, dy).to(weight.dtype)
        else:
            link = ctx.res_link if not has_res else None
            # the residual gradient conv2's backward left / the gradient the alias' other consumer left (fan link)
            into = link.usable((n, cin, h, w), T) if link is not None else None

            def dgrad():
                cl = link.chunk if link is not None else None
                dst = cl.dst if cl is not None else None
                if dst is not None and stride == 1 and tuple(dst.shape) == (n, cin, h, w) and dst.dtype == T and ops.is_nhwc(dst):
                    wb = ops.pack_weights(weight, k, stride, 1, T)
                    if link.fan:                            # C3K entry: both 1x1 convs add into the concat's slice
                        ops.conv_dgrad(dy, wb, cin, h, w, k, stride, acc_into=into if into is not None else dst)
                        if into is None:
                            link.dres = dst
                        cl.done = True
                        return None
                    if into is not None:                    # Residual entry: slice += skip gradient + this data gradient
                        ops.conv_dgrad(dy, wb, cin, h, w, k, stride, acc_into=dst, acc2=into)
                        cl.done = True
                        return None
                r = ops.conv_dgrad(dy, ops.pack_weights(weight, k, stride, 1, T), cin, h, w, k, stride, acc_into=into)
                if into is not None:
                    return None                             # already inside the gradient autograd holds for x
                if link is not None and link.fan:
                    link.dres = r                           # first of the two consumers: the other adds to this
                return r
            if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
                dx, dw = _wgrad_overlapped(x, dy, k, stride, weight.dtype, dgrad)
            elif ctx.needs_input_grad[0]:
                dx = dgrad()
            elif ctx.needs_input_grad[1]:
                dw = ops.conv_wgrad(x, dy, k, stride, weight.dtype)
        if dgamma is None or not ctx.needs_input_grad[2]:      # (the kernels wrote them in the parameters' own dtype)
            dgamma = dbeta = None
        dres = dout if (has_res and ctx.needs_input_grad[4]) else None
        if has_res and ctx.res_link is not None and not ctx.res_link.fan:
            ctx.res_link.dres = dres
        return dx, dw, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None


class ConvBias(torch.autograd.Function):
    """Plain dense conv + bias: the head's final nn.Conv2d 1x1 (src/model/head.py:50,60)."""

    @staticmethod
    def forward(ctx, *args):
        if FWD_STREAM is not None:
            with torch.cuda.stream(FWD_STREAM):
                return ConvBias._forward(ctx, *args)
        return ConvBias._forward(ctx, *args)

    @staticmethod
    def _forward(ctx, x, weight, bias, k, stride):
        T = compute_dtype(x, weight)
        x = _as_nhwc(x, T)
        cout = weight.shape[0]
        b32 = _f32(bias) if bias is not None else None
        y = ops.conv_fwd(x, ops.pack_weights(weight, k, stride, 0, T), b32, cout, k, stride)
        ctx.cfg = (k, stride, tuple(x.shape), None if bias is None else bias.dtype)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        k, stride, xshape, bdtype = ctx.cfg
        x, weight = ctx.saved_tensors
        dy = _as_nhwc(dy, x.dtype)
        n, cin, h, w = xshape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_dgrad(dy, ops.pack_weights(weight, k, stride, 1, x.dtype), cin, h, w, k, stride)
        lazy = x.is_cuda and LAZY_WGRAD_JOIN
        if ctx.needs_input_grad[1]:
            if lazy:
                dwb = torch.empty(weight.shape, dtype=weight.dtype, device=x.device)
                fn = lambda: ops.conv_wgrad(x, dy, k, stride, dwb.dtype, out=dwb)
                if not defer_to_side(x.device, (x, dy), fn):
                    fn()
                dw = dwb.detach()
            else:
                dw = ops.conv_wgrad(x, dy, k, stride, weight.dtype)
        if bdtype is not None and ctx.needs_input_grad[2]:
            if lazy and bdtype == torch.float32:
                dbb = torch.empty(weight.shape[0], dtype=torch.float32, device=x.device)
                fnb = lambda: ops.channel_sum(dy, out=dbb)
                if not defer_to_side(x.device, (dy,), fnb):
                    fnb()
                db = dbb.detach()
            else:
                db = ops.channel_sum(dy).to(bdtype)
        return dx, dw, db, None, None


def _frozen_packed(weight, k, stride, T):
    """Forward-packed matrix of a FROZEN weight (a fused conv's: requires_grad False), packed once and kept on the parameter
    until it is written to (version counter) or computed in another dtype."""
    hit = getattr(weight, "_yolo_packed", None)
    if hit is not None and hit[0] == (weight._version, weight.data_ptr(), T):
        return hit[1]
    wp = ops.pack_weights(weight, k, stride, 0, T)
    if not weight.requires_grad:
        weight._yolo_packed = ((weight._version, weight.data_ptr(), T), wp)
    return wp


def _frozen_stem(weight, T):
    """[cout][32] stem matrix of a frozen weight, kept on the parameter like _frozen_packed."""
    hit = getattr(weight, "_yolo_stem_packed", None)
    if hit is not None and hit[0] == (weight._version, weight.data_ptr(), T):
        return hit[1]
    wp = ops.stem_pack_weights(weight, T)
    if not weight.requires_grad:
        weight._yolo_stem_packed = ((weight._version, weight.data_ptr(), T), wp)
    return wp


def fused_conv_act(x, weight, bias, k, stride, depthwise, act, res=None, out=None):
    """Conv with BN folded in (Model.fuse(), src/model/model_blocks.py:36-37): act(conv(x) + b) (+ res).
    Inference-only like the reference's fused conv (requires_grad False); no autograd node."""
    with torch.no_grad():
        T = compute_dtype(x, weight)
        cout = weight.shape[0]
        b32 = _f32(bias)
        if (not depthwise and weight.shape[1] == 3 and k == 3 and stride == 2 and res is None and out is None and x.is_cuda
                and x.dtype == torch.float32 and ops.stem_conv_eligible(x, T, cout)):
            # the stem: straight from the NCHW fp32 image, bias + SiLU in the epilogue (Cin = 3 has no MFMA conv kernel)
            return ops.stem_conv_fwd(x, _frozen_stem(weight, T), cout, T, None, b32, act)
        x = _as_nhwc(x, T)
        if not depthwise and act == ACT_IDENTITY and res is None:
            return ops.conv_fwd(x, _frozen_packed(weight, k, stride, T), b32, cout, k, stride, out=out)
        res = None if res is None else _as_nhwc(res, T)
        if depthwise:
            w9 = _f32(weight).reshape(cout, 9)
            if res is None and out is None and T in _LOWP and x.is_cuda:
                y = ops.dw_fwd_act(x, w9, b32, act)
                if y is not None:
                    return y
            y = ops.dw_fwd(x, w9)
        else:
            # bias + SiLU + residual in the conv's own epilogue: one launch, one pass (16-bit MFMA path)
            wp = _frozen_packed(weight, k, stride, T)
            y = ops.conv_fwd_act(x, wp, b32, cout, k, stride, act, res, out) if T in _LOWP and x.is_cuda else None
            if y is not None:
                return y
            y = ops.conv_fwd(x, wp, None, cout, k, stride)
        one = ops.fill_(torch.empty(cout, dtype=torch.float32, device=x.device), 1.0)
        return ops.bn_act_fwd(y, one, b32, act, res, out)


class Cat(torch.autograd.Function):
    """torch.cat(dim=1) as channel-slice copies into one NHWC buffer; backward hands out slices."""

    @staticmethod
    def forward(ctx, *xs):
        T = xs[0].dtype
        xs = [_as_nhwc(x, T) for x in xs]
        n, _, h, w = xs[0].shape
        cs = [x.shape[1] for x in xs]
        out = ops.new_nhwc(n, sum(cs), h, w, T, xs[0].device)
        off = 0
        for x, c in zip(xs, cs):
            ops.copy_channels(x, out[:, off:off + c])
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _as_nhwc(dout, dout.dtype)
        grads, off = [], 0
        for c in ctx.cs:
            grads.append(dout[:, off:off + c])
            off += c
        return tuple(grads)


def _fan_bwd(link, shape, dtype, run):
    """Gradient of one of the two consumers of a fan2 alias: `run(acc_into)` computes it, added to the other consumer's
    gradient when that one came first (returns None: nothing for autograd to add), else left in the link."""
    if link is None:
        return run(None)
    into = link.usable(shape, dtype)
    g = run(into)
    if into is not None:
        return None
    link.dres = g
    return g


def _fresh(t):
    """A Function output must not BE one of its inputs: hand back a new tensor object on the same memory
    (the caller's `out=` view) so autograd attaches the node to it without view / in-place bookkeeping."""
    return t.detach() if t is not None else t


def cat_buffer(like, weight, channels):
    """Concat buffer for producers that write their channel slice in place (see CatInto): NHWC, the compute dtype
    the producers will use for `like` (autocast-aware), `like`'s batch and map size."""
    n, _, h, w = like.shape
    return ops.new_nhwc(n, channels, h, w, compute_dtype(like, weight), like.device)


class CatInto(torch.autograd.Function):
    """torch.cat(dim=1) without the copies: `buf` is the concat buffer, xs[i] its consecutive channel slices.  A
    producer that was handed its slice as `out=` has already written it (same memory: nothing to do); any other
    input (a tensor produced elsewhere, e.g. a backbone feature entering the neck) is copied in.  Backward hands
    out slices of the incoming gradient, like Cat."""

    pending = None          # (ChunkLink, index of the part whose gradient slice it wants): set by cat_into, taken by forward

    @staticmethod
    def forward(ctx, buf, *xs):
        ctx.link, CatInto.pending = CatInto.pending, None
        off, cs = 0, []
        for x in xs:
            c = x.shape[1]
            dst = buf[:, off:off + c]
            if x.data_ptr() != dst.data_ptr() or x.stride() != dst.stride():
                ops.copy_channels(_as_nhwc(x, buf.dtype), dst)
            cs.append(c)
            off += c
        if off != buf.shape[1]:
            raise RuntimeError("CatInto: slices do not cover the buffer")
        ctx.cs = cs
        return _fresh(buf)

    @staticmethod
    def backward(ctx, dout):
        dout = _as_nhwc(dout, dout.dtype)
        grads, off = [None], 0
        for c in ctx.cs:
            grads.append(dout[:, off:off + c])
            off += c
        if ctx.link is not None:
            cl, index = ctx.link
            cl.dst, cl.done = grads[1 + index], False
        return tuple(grads)


def cat_into(buf, parts, chunk_link=None, index=1):
    """CatInto.apply with a ChunkLink that wants the gradient slice of parts[index]."""
    CatInto.pending = (chunk_link, index) if chunk_link is not None else None
    return CatInto.apply(buf, *parts)


class Chunk2(torch.autograd.Function):
    """x.chunk(2, 1): two channel-slice views; backward re-assembles the halves (zeros where unused).
    `fanout=True` returns the second half twice (for a caller that feeds it to two consumers, C3K2: the concat and
    the first block): autograd then delivers the two gradients separately instead of adding them first, and when the
    gradients of both halves are neighbouring slices of one buffer (the concat's gradient) the third is accumulated
    into that buffer in place and the buffer's head IS the result -- one launch instead of an add and two copies."""

    @staticmethod
    def forward(ctx, x, fanout=False, chunk_link=None):
        x = _as_nhwc(x, x.dtype)
        h = x.shape[1] // 2
        ctx.shape = tuple(x.shape)
        ctx.chunk_link = chunk_link
        if fanout:
            return x[:, :h], x[:, h:], x[:, h:]
        return x[:, :h], x[:, h:]

    @staticmethod
    def backward(ctx, g0, g1, g2=None):
        n, c, h, w = ctx.shape
        half = c // 2
        cl = ctx.chunk_link
        if cl is not None:
            if cl.done:
                g2 = None                                   # the block's entry convs already added it into g1 (ChunkLink)
            cl.dst, cl.done = None, False
        if g0 is not None and g1 is not None and g0.dtype == g1.dtype and g0.stride() == g1.stride() \
                and ops.is_nhwc(g0) and ops.geom(g0)[4] >= c \
                and g1.data_ptr() == g0.data_ptr() + half * g0.element_size():
            if g2 is not None:
                ops.copy_channels(_as_nhwc(g2, g1.dtype), g1, accumulate=True)
            return g0.as_strided((n, c, h, w), g0.stride(), g0.storage_offset()), None, None
        ref = g0 if g0 is not None else (g1 if g1 is not None else g2)
        dx = ops.new_nhwc(n, c, h, w, ref.dtype, ref.device)
        for g, sl in ((g0, dx[:, :half]), (g1, dx[:, half:])):
            if g is None:
                ops.copy_channels(ops.zero_(ops.new_nhwc(n, sl.shape[1], h, w, ref.dtype, ref.device)), sl)
            else:
                ops.copy_channels(_as_nhwc(g, ref.dtype), sl)
        if g2 is not None:
            ops.copy_channels(_as_nhwc(g2, ref.dtype), dx[:, half:], accumulate=True)
        return dx, None, None


class MaxPool5(torch.autograd.Function):
    """nn.MaxPool2d(5, 1, 2) (src/model/model_blocks.py:150)."""

    @staticmethod
    def forward(ctx, x, out=None, link=None):
        out, idx = ops.maxpool5_fwd(_as_nhwc(x, x.dtype), out)
        ctx.save_for_backward(idx)
        ctx.link = link
        return _fresh(out)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = _as_nhwc(dout, dout.dtype)
        return _fan_bwd(ctx.link, dout.shape, dout.dtype, lambda into: ops.maxpool5_bwd(dout, idx, acc_into=into)), None, None


class Upsample2x(torch.autograd.Function):
    """nn.Upsample(scale_factor=2), nearest (src/model/neck.py:31)."""

    @staticmethod
    def forward(ctx, x, out=None, link=None):
        ctx.link = link
        return _fresh(ops.upsample2x_fwd(_as_nhwc(x, x.dtype), out))

    @staticmethod
    def backward(ctx, dout):
        dout = _as_nhwc(dout, dout.dtype)
        n, c, oh, ow = dout.shape
        return _fan_bwd(ctx.link, (n, c, oh // 2, ow // 2), dout.dtype, lambda into: ops.upsample2x_bwd(dout, acc_into=into)), None, None


class AttentionCore(torch.autograd.Function):
    """softmax(q^T k * scale) applied to v, per head, plus v re-gathered for the positional dw-conv
    (src/model/model_blocks.py:190-197).  qkv is the NHWC output of the qkv Conv."""

    @staticmethod
    def forward(ctx, qkv, heads, dk, dh, scale):
        qkv = _as_nhwc(qkv, qkv.dtype)
        if not ctx.needs_input_grad[0] and qkv.is_cuda:       # inference / no-grad: nothing to keep, any sequence length
            got = ops.attn_fwd_nograd(qkv, heads, dk, dh, scale)
            if got is not None:
                return got
        o, vp, lse = ops.attn_fwd(qkv, heads, dk, dh, scale)
        ctx.cfg = (heads, dk, dh, scale)
        ctx.save_for_backward(qkv, o, lse)
        return o, vp

    @staticmethod
    def backward(ctx, d_o, d_vp):
        heads, dk, dh, scale = ctx.cfg
        qkv, o, lse = ctx.saved_tensors
        T = qkv.dtype
        if d_o is None:
            d_o = ops.zero_(ops.new_nhwc(*o.shape, T, o.device))
        d_o = _as_nhwc(d_o, T)
        d_vp = _as_nhwc(d_vp, T) if d_vp is not None else None
        return ops.attn_bwd(qkv, o, d_o, d_vp, lse, heads, dk, dh, scale), None, None, None, None


class HeadPack(torch.autograd.Function):
    """cat(box, cls) per level -> view(N, no, -1) -> cat over levels (src/model/head.py:87,119):
    NHWC branch outputs are transposed straight into the (N, 64+nc, M) prediction tensor."""

    @staticmethod
    def forward(ctx, *branches):
        T = branches[0].dtype
        branches = [_as_nhwc(b, T) for b in branches]
        n = branches[0].shape[0]
        levels = [(branches[i], branches[i + 1]) for i in range(0, len(branches), 2)]
        no = levels[0][0].shape[1] + levels[0][1].shape[1]
        m = sum(b.shape[2] * b.shape[3] for b, _ in levels)
        preds = torch.empty((n, no, m), dtype=T, device=branches[0].device)
        meta, m_off, c_offs, m_offs = [], 0, [], []
        for box, cls in levels:
            h, w = box.shape[2], box.shape[3]
            c_offs += [0, box.shape[1]]
            m_offs += [m_off, m_off]
            meta.append((box.shape[1], cls.shape[1], h, w, m_off))
            m_off += h * w
        ops.head_group(branches, preds, c_offs, m_offs, True)             # all branches in one launch
        ctx.meta = meta
        return preds

    @staticmethod
    def backward(ctx, dpreds):
        dpreds = dpreds.contiguous()
        n = dpreds.shape[0]
        grads, c_offs, m_offs = [], [], []
        for cb, cc, h, w, m_off in ctx.meta:
            grads += [ops.new_nhwc(n, cb, h, w, dpreds.dtype, dpreds.device), ops.new_nhwc(n, cc, h, w, dpreds.dtype, dpreds.device)]
            c_offs += [0, cb]
            m_offs += [m_off, m_off]
        ops.head_group(grads, dpreds, c_offs, m_offs, False)
        return tuple(grads)


# Set by TrainStepRunner around `loss.backward()` when `loss` IS this node's output: autograd seeds it with ones, so the
# incoming gradient is exactly 1.0 and the in-place scale of dpreds (77 MB read + written, 30 us at the head of the backward
# chain) is the identity.  Everywhere else the gradient is applied as it comes (GradScaler, weighted sums of losses).
UNIT_LOSS_SEED = False
# fp16 loss scaling on the device (TrainStepRunner with precision float16): an fp32 device scalar the loss kernel multiplies
# its gradient by IN FP32, before the one rounding to fp16 -- `scaler.scale(loss).backward()` of the reference
# (src/training/train_model.py:247-253), where the scaled gradient is formed in fp32 and cast at `preds.float()`.  The
# loss VALUE stays unscaled.  None everywhere else (a torch GradScaler's factor then arrives as the incoming gradient).
LOSS_SCALE = None


class DflQflLoss(torch.autograd.Function):
    """YoloDFLQFLoss value and gradient from one fused pass (src/model/losses.py:140-281).
    Returns (total, scalars[3]); scalars = (total, mean_dfl, mean_cls) detached."""

    @staticmethod
    def forward(ctx, preds, anchors, strides, packed, nc, lambda_dfl, lambda_cls):
        gt, gt_off, gt_img, n_gt = packed
        want = ctx.needs_input_grad[0]
        out, dpreds, _ = ops.loss_fwd_bwd(preds.contiguous(), anchors, strides, gt, gt_off, gt_img, n_gt, nc,
                                          lambda_dfl, lambda_cls, want, LOSS_SCALE if want else None)
        ctx.dpreds = dpreds
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_scalars):
        dpreds = ctx.dpreds
        ctx.dpreds = None
        if dpreds is None:
            return (None,) * 7
        if not UNIT_LOSS_SEED:      # a seed of exactly 1.0 (see UNIT_LOSS_SEED) leaves every 16-bit value as it is: no pass over dpreds
            ops.scale_inplace(dpreds, g_total.reshape(1).float())
        return dpreds, None, None, None, None, None, None
