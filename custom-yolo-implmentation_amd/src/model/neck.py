"""PAN-FPN neck: top-down (x2 nearest upsample + concat + C3K2) twice, bottom-up (stride-2 Conv +
concat + C3K2) twice.  Mirrors the reference's src/model/neck.py:31-45 (h1..h6 are state-dict keys)."""
from typing import List

from torch import nn

from src.hipops import functions as F_
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

    def forward(self, x):
        p3, p4, p5 = x
        up, cat = F_.Upsample2x.apply, F_.Cat.apply
        p4 = self.h1(cat(up(p5), p4))
        p3 = self.h2(cat(up(p4), p3))
        p4 = self.h4(cat(self.h3(p3), p4))
        p5 = self.h6(cat(self.h5(p4), p5))
        return p3, p4, p5
