"""Building blocks of the detector, MI355X-native.

Same classes, constructor arguments, attribute names and state-dict keys as the reference's
src/model/model_blocks.py (so checkpoints, DDP/FSDP wrapping by class and `Model.fuse()` behave the
same), but `forward` never touches an ATen compute op: every block runs on the HIP kernels of
libyolo_hip.so through src.hipops.functions, on NHWC bf16/f16/fp32 activations.  nn.Conv2d /
nn.BatchNorm2d objects are kept purely as parameter containers (names + default initialisation).
"""
import torch
from torch import nn

from src.hipops import functions as F_
from src.hipops.lib import ACT_IDENTITY, ACT_SILU


def _act_code(m: nn.Module) -> int:
    if isinstance(m, nn.SiLU):
        return ACT_SILU
    if isinstance(m, nn.Identity):
        return ACT_IDENTITY
    raise ValueError(f"activation {type(m).__name__} has no HIP epilogue (SiLU and Identity do)")


class Conv(nn.Module):
    """conv(bias=False) -> BatchNorm2d(eps=1e-3, momentum=0.03) -> activation, one fused unit.
    Reference: src/model/model_blocks.py:4-37.  `forward(x, residual)` adds `residual` after the
    activation in the same kernel (used by Residual / PSABlock)."""

    def __init__(self, in_ch: int, out_ch: int, activation: nn.Module, k: int = 1, s: int = 1, p: int = 0, g: int = 1):
        super().__init__()
        if p != k // 2 or (g != 1 and not (g == in_ch == out_ch and k == 3 and s == 1)) \
                or (k, s) not in ((1, 1), (3, 1), (3, 2)):
            raise ValueError(f"Conv(k={k}, s={s}, p={p}, g={g}) is outside the shapes the HIP kernels cover")
        self.conv = nn.Conv2d(in_ch, out_ch, k, s, p, groups=g, bias=False)
        self.norm = nn.BatchNorm2d(out_ch, eps=0.001, momentum=0.03)
        self.relu = activation
        self._k, self._s, self._dw, self._act = k, s, g != 1, _act_code(activation)
        self._count_batches = True      # a parent Model bumps all counters in one multi-tensor op instead

    def forward(self, x, residual=None, out=None, res_link=None):
        """`out`: optional destination view (a channel slice of the consumer's concat buffer, F_.cat_buffer).
        `res_link`: see F_.ResLink (Residual)."""
        n = self.norm
        if self.training and self._count_batches:
            n.num_batches_tracked.add_(1)
        return F_.ConvBnAct.apply(x, self.conv.weight, n.weight, n.bias, residual, (n.running_mean, n.running_var),
                                  self._k, self._s, self._dw, self._act, self.training, n.momentum, n.eps, out, res_link)

    def fuse_forward(self, x, residual=None, out=None, res_link=None):
        return F_.fused_conv_act(x, self.conv.weight, self.conv.bias, self._k, self._s, self._dw, self._act, residual, out)


class Residual(nn.Module):
    """x + Conv3x3(Conv3x3(x)); reference :39-62."""

    def __init__(self, ch: int, e: float = 0.5):
        super().__init__()
        mid = int(ch * e)
        self.conv1 = Conv(ch, mid, nn.SiLU(), k=3, p=1)
        self.conv2 = Conv(mid, ch, nn.SiLU(), k=3, p=1)

    def forward(self, x, out=None, chunk_link=None):
        if not (self.training and torch.is_grad_enabled()):
            return self.conv2(self.conv1(x), x, out=out)
        # conv1's data gradient is accumulated into the residual gradient by its own kernel (F_.ResLink); the alias
        # makes sure the two convs are the only consumers of what they see as x
        xp, link = F_.Alias.apply(x), F_.ResLink()
        link.chunk = chunk_link
        return self.conv2(self.conv1(xp, res_link=link), xp, out=out, res_link=link)


class C3K(nn.Module):
    """Two 1x1 branches, two Residual(e=1) on the first, concat, 1x1; reference :64-92."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        half = out_ch // 2
        self.conv1 = Conv(in_ch, half, nn.SiLU())
        self.conv2 = Conv(in_ch, half, nn.SiLU())
        self.conv3 = Conv(2 * half, out_ch, nn.SiLU())
        self.res_m = nn.Sequential(Residual(half, e=1.0), Residual(half, e=1.0))

    def _fused_pair(self):
        """After Model.fuse(): the two entry 1x1 convs read the same x, so they are ONE conv with the weights stacked along the
        output channels (frozen: stacked once, kept until a weight is written to or moved) -- one launch and one read of x
        instead of two."""
        c1, c2 = self.conv1.conv, self.conv2.conv
        key = (c1.weight.data_ptr(), c2.weight.data_ptr(), c1.weight._version, c2.weight._version, c1.bias._version, c2.bias._version)
        hit = self.__dict__.get("_pair")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w = torch.cat([c1.weight, c2.weight], 0).contiguous()
                b = torch.cat([c1.bias, c2.bias], 0).contiguous()
            hit = self.__dict__["_pair"] = (key, w, b)
        return hit[1], hit[2]

    def forward(self, x, out=None, chunk_link=None):
        # both branches write their half of the concat buffer directly (no torch.cat copy)
        half = self.conv1.conv.out_channels
        buf = F_.cat_buffer(x, self.conv1.conv.weight, 2 * half)
        if not hasattr(self.conv1, "norm") and not hasattr(self.conv2, "norm") and not torch.is_grad_enabled() and x.is_cuda:
            # fused inference: the stacked conv fills the whole buffer; the first half is the residual chain's input and is
            # overwritten by that chain's last block once its first block has consumed it (stream order)
            w, b = self._fused_pair()
            ab = F_.fused_conv_act(x, w, b, 1, 1, False, self.conv1._act, None, buf)
            a = self.res_m[1](self.res_m[0](ab[:, :half]), out=buf[:, :half])
            return self.conv3(F_.CatInto.apply(buf, a, ab[:, half:]), out=out)
        xa, link = F_.fan2(x)                                 # two 1x1 convs read x: the second data gradient is added to the first
        if link is not None:
            link.chunk = chunk_link
        a = self.res_m[1](self.res_m[0](self.conv1(xa, res_link=link)), out=buf[:, :half])
        b = self.conv2(xa, out=buf[:, half:], res_link=link)
        return self.conv3(F_.CatInto.apply(buf, a, b), out=out)


class C3K2(nn.Module):
    """1x1 -> split in two -> n chained (Residual | C3K) on the newest piece -> concat all -> 1x1;
    reference :94-125."""

    def __init__(self, in_ch: int, out_ch: int, n: int, csp: bool, r: int):
        super().__init__()
        h = out_ch // r
        self.conv1 = Conv(in_ch, 2 * h, nn.SiLU())
        self.conv2 = Conv((2 + n) * h, out_ch, nn.SiLU())
        self.res_m = nn.ModuleList((C3K(h, h) if csp else Residual(h)) for _ in range(n))

    def forward(self, x, out=None):
        # conv1 and every chained block write straight into the concat buffer conv2 reads
        h = self.conv1.conv.out_channels // 2
        buf = F_.cat_buffer(x, self.conv1.conv.weight, (2 + len(self.res_m)) * h)
        # two views of the buffer's head; the second half also feeds the first block (see Chunk2 on fan-out)
        # ... and the first block's entry convs add their data gradients straight into the concat gradient's slice of b
        cl = F_.ChunkLink() if (self.training and torch.is_grad_enabled() and F_.CHUNK_LINK) else None
        a, b, b_blk = F_.Chunk2.apply(self.conv1(x, out=buf[:, :2 * h]), True, cl)
        parts = [a, b]
        for i, m in enumerate(self.res_m):
            o = buf[:, (2 + i) * h:(3 + i) * h]
            parts.append(m(b_blk, out=o, chunk_link=cl) if i == 0 else m(parts[-1], out=o))
        return self.conv2(F_.cat_into(buf, parts, cl, 1), out=out)


class SPPF(nn.Module):
    """1x1, three chained 5x5 max pools, concat of the four maps, 1x1; reference :127-156."""

    def __init__(self, c1: int, c2: int, k: int = 5):
        super().__init__()
        if k != 5:
            raise ValueError("the HIP pooling kernel is 5x5 (the only size the model uses)")
        self.cv1 = Conv(c1, c1 // 2, nn.SiLU(), 1, 1)
        self.cv2 = Conv((c1 // 2) * 4, c2, nn.SiLU(), 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)   # kept for module-tree parity

    def forward(self, x):
        c = self.cv1.conv.out_channels
        buf = F_.cat_buffer(x, self.cv1.conv.weight, 4 * c)
        # each map feeds the concat and the next pool: the pool's gradient is added to the concat slice's (F_.fan2)
        x, l0 = F_.fan2(self.cv1(x, out=buf[:, :c]))
        y1, l1 = F_.fan2(F_.MaxPool5.apply(x, buf[:, c:2 * c], l0))
        y2, l2 = F_.fan2(F_.MaxPool5.apply(y1, buf[:, 2 * c:3 * c], l1))
        y3 = F_.MaxPool5.apply(y2, buf[:, 3 * c:], l2)
        return self.cv2(F_.CatInto.apply(buf, F_.stash(x, l0), F_.stash(y1, l1), F_.stash(y2, l2), y3))


class Attention(nn.Module):
    """Multi-head attention over the H*W tokens of the stride-32 map with a depthwise positional
    term; reference :158-198."""

    def __init__(self, ch: int, num_head: int):
        super().__init__()
        self.num_head = num_head
        self.dim_head = ch // num_head
        self.dim_key = self.dim_head // 2
        self.scale = self.dim_key ** -0.5
        self.qkv = Conv(ch, ch + self.dim_key * num_head * 2, nn.Identity())
        self.conv1 = Conv(ch, ch, nn.Identity(), k=3, p=1, g=ch)
        self.conv2 = Conv(ch, ch, nn.Identity())

    def forward(self, x, residual=None, res_link=None):
        """`res_link` (F_.ResLink, with residual = the SAME alias as x): the qkv conv's data gradient is accumulated
        into the residual gradient conv2's backward leaves in the link."""
        o, v = F_.AttentionCore.apply(self.qkv(x, res_link=res_link), self.num_head, self.dim_key, self.dim_head, self.scale)
        return self.conv2(self.conv1(v, o), residual, res_link=res_link)        # conv2(attn_out + dwconv(v)) (+ x)


class PSABlock(nn.Module):
    """x + Attention(x), then x + FFN(x); reference :200-224."""

    def __init__(self, ch: int, num_head: int):
        super().__init__()
        self.conv1 = Attention(ch, num_head)
        self.conv2 = nn.Sequential(Conv(ch, ch * 2, nn.SiLU()), Conv(ch * 2, ch, nn.Identity()))

    def forward(self, x):
        if not (self.training and torch.is_grad_enabled()):
            x = self.conv1(x, x)
            return self.conv2[1](self.conv2[0](x), x)
        # both residual sums as in Residual: the first consumer's data gradient is accumulated into the residual
        # gradient by its own kernel epilogue (F_.ResLink) instead of an extra add pass
        xp, link = F_.Alias.apply(x), F_.ResLink()
        x = self.conv1(xp, xp, res_link=link)
        xp, link = F_.Alias.apply(x), F_.ResLink()
        return self.conv2[1](self.conv2[0](xp, res_link=link), xp, res_link=link)


class PSA(nn.Module):
    """1x1, split, n PSABlocks on the second half, concat, 1x1; reference :226-252."""

    def __init__(self, ch: int, n: int):
        super().__init__()
        self.conv1 = Conv(ch, 2 * (ch // 2), nn.SiLU())
        self.conv2 = Conv(2 * (ch // 2), ch, nn.SiLU())
        self.res_m = nn.Sequential(*(PSABlock(ch // 2, ch // 128) for _ in range(n)))

    def forward(self, x, out=None):
        a, b = F_.Chunk2.apply(self.conv1(x), False)
        return self.conv2(F_.Cat.apply(a, self.res_m(b)), out=out)


class DFL(nn.Module):
    """Softmax-expectation over the 16 distance bins; frozen 1x1 conv weight 0..c1-1 kept as a
    parameter for state-dict parity (reference :254-280).  Model.inference uses the fused
    yolo_head_decode kernel; this forward (same math, yolo_dfl_expect) keeps the module API."""

    def __init__(self, c1: int = 16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        from src.hipops import ops
        if x.shape[1] != 4 * self.c1 or self.c1 != 16:
            raise ValueError("DFL kernel is built for 4 x 16 bins")
        return ops.dfl_expect(x)
