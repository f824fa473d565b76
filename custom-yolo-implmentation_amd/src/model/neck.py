"""PAN-FPN neck: top-down (x2 nearest upsample + concat + C3K2) twice, bottom-up (stride-2 Conv +
concat + C3K2) twice.  Mirrors the reference's src/model/neck.py:31-45 (h1..h6 are state-dict keys)."""
from typing import List

from torch import nn

from src.hipops import functions as F_
from src.hipops import ops
from src.model.model_blocks import C3K2, Conv


class Neck(nn.Module):
    def __init__(self, width: List[int], depth: List[int], csp: List[bool]):
        super().__init__()
        w, n = width, depth[5]
        self.up = nn.Upsample(scale_factor=2)          # module-tree parity; the HIP kernel does the work
        self.h1 = C3K2(w[4] + w[5], w[4], n, csp[0], r=2)
        self.h2 = C3K2(w[4] + w[4], w[3], n, csp[0], r=2)
        self.h3 = Conv(w[3], w[3], nn.SiLU(), k=3, s=2, p=1)
        self.h4 = C3K2(w[3] + w[4], w[4], n, csp[0], r=2)
        self.h5 = Conv(w[4], w[4], nn.SiLU(), k=3, s=2, p=1)
        self.h6 = C3K2(w[4] + w[5], w[5], n, csp[1], r=2)

    def alloc(self, x):
        """The four concat buffers for an input batch x (N, 3, H, W), and the slices the backbone's p3 / p4 / p5 outputs
        belong in: (bufs, (o3, o4, o5)).  Handed to Backbone.forward / Neck.forward by Model.forward."""
        n, _, h, w = x.shape
        hw = []
        for _ in range(5):
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            hw.append((h, w))
        (h8, w8), (h16, w16), (h32, w32) = hw[2], hw[3], hw[4]
        c3, c4, c5 = self.h3.conv.in_channels, self.h5.conv.in_channels, self.h6.conv2.conv.out_channels
        T = F_.compute_dtype(x, self.h1.conv1.conv.weight)
        new = lambda c, hh, ww: ops.new_nhwc(n, c, hh, ww, T, x.device)
        bufs = dict(h1=new(c5 + c4, h16, w16), h2=new(c4 + c4, h8, w8), h4=new(c3 + c4, h16, w16), h6=new(c4 + c5, h32, w32))
        return bufs, (bufs["h2"][:, c4:], bufs["h1"][:, c5:], bufs["h6"][:, c4:])

    def forward(self, x, bufs=None):
        p3, p4, p5 = x
        bufs = bufs or {}

        def joined(head, make_first, skip, buf=None, out=None):
            """cat(first, skip) for `head`: `first` (an upsample / stride-2 Conv output) is written straight into the
            concat buffer; `skip` already sits in its slice when its producer was handed that slice (Neck.alloc), else
            CatInto copies it in.  `out`: the slice of a LATER concat buffer this block's own output belongs in."""
            c1 = make_first[1]
            if buf is None or buf.shape[1] != c1 + skip.shape[1] or buf.shape[2:] != skip.shape[2:] or buf.shape[0] != skip.shape[0]:
                buf = F_.cat_buffer(skip, head.conv1.conv.weight, c1 + skip.shape[1])
            first = make_first[0](buf[:, :c1])
            cat = F_.CatInto.apply(buf, first, skip)
            return head(cat, out=out) if out is not None else head(cat)

        c3 = self.h3.conv.out_channels
        o4 = bufs["h4"][:, c3:] if "h4" in bufs else None            # h1's output = the skip of h4's concat
        # every tensor here has two consumers; F_.fan2: the second gradient is added to the first by the kernel that
        # produces it (upsample backward, stride-2 conv data gradient), autograd never sums
        p5, l5 = F_.fan2(p5)
        p4, l4 = F_.fan2(joined(self.h1, (lambda o: F_.Upsample2x.apply(p5, o, l5), p5.shape[1]), p4, bufs.get("h1"), o4))
        p3, l3 = F_.fan2(joined(self.h2, (lambda o: F_.Upsample2x.apply(p4, o, l4), p4.shape[1]), p3, bufs.get("h2")))
        p4b, l4b = F_.fan2(joined(self.h4, (lambda o: self.h3(p3, out=o, res_link=l3), self.h3.conv.out_channels), F_.stash(p4, l4),
                                  bufs.get("h4")))
        p5b = joined(self.h6, (lambda o: self.h5(p4b, out=o, res_link=l4b), self.h5.conv.out_channels), F_.stash(p5, l5),
                     bufs.get("h6"))
        return F_.stash(p3, l3), F_.stash(p4b, l4b), p5b
