"""Feature extractor: five stride-2 stages, C3K2 after each from p2 on, SPPF + PSA on p5.
Mirrors the reference's src/model/backbone.py:37-66 (stage names p1..p5 are state-dict keys)."""
from typing import List

from torch import nn

from src.hipops import functions as F_
from src.model.model_blocks import C3K2, PSA, SPPF, Conv


def _down(cin, cout):
    return Conv(cin, cout, nn.SiLU(), k=3, s=2, p=1)


class Backbone(nn.Module):
    def __init__(self, width: List[int], depth: List[int], csp: List[bool]):
        super().__init__()
        w = width
        self.p1 = nn.Sequential(_down(w[0], w[1]))
        self.p2 = nn.Sequential(_down(w[1], w[2]), C3K2(w[2], w[3], depth[0], csp[0], r=4))
        self.p3 = nn.Sequential(_down(w[3], w[3]), C3K2(w[3], w[4], depth[1], csp[0], r=4))
        self.p4 = nn.Sequential(_down(w[4], w[4]), C3K2(w[4], w[4], depth[2], csp[1], r=2))
        self.p5 = nn.Sequential(_down(w[4], w[5]), C3K2(w[5], w[5], depth[3], csp[1], r=2),
                                SPPF(w[5], w[5]), PSA(w[5], depth[4]))

    def forward(self, x, outs=None):
        """`outs` = (o3, o4, o5): channel slices of the neck's concat buffers (Neck.alloc) that the last conv of stages
        p3 / p4 / p5 writes straight into, so that the neck's torch.cat needs no copy of a backbone feature."""
        o3, o4, o5 = outs if outs is not None else (None, None, None)
        # p3 / p4 feed the next stage (a stride-2 Conv) and the neck: that conv's data gradient is ADDED to the neck's
        # gradient in its kernel epilogue (F_.fan2) instead of autograd summing the two
        p3, l3 = F_.fan2(self._stage(self.p3, self.p2(self.p1(x)), None, o3))
        p4, l4 = F_.fan2(self._stage(self.p4, p3, l3, o4))
        return F_.stash(p3, l3), F_.stash(p4, l4), self._stage(self.p5, p4, l4, o5)

    @staticmethod
    def _stage(seq, x, link, out=None):
        x = seq[0](x, res_link=link)
        for m in seq[1:-1]:
            x = m(x)
        return seq[-1](x, out=out) if out is not None else seq[-1](x)
