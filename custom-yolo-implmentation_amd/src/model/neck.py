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

        def joined(head, make_first, skip):
            """cat(first, skip) for `head`: `first` (an upsample / stride-2 Conv output) is written straight into the
            concat buffer; `skip` comes from another stage and is copied in by CatInto."""
            c1 = make_first[1]
            buf = F_.cat_buffer(skip, head.conv1.conv.weight, c1 + skip.shape[1])
            first = make_first[0](buf[:, :c1])
            return head(F_.CatInto.apply(buf, first, skip))

        # every tensor here has two consumers; F_.fan2: the second gradient is added to the first by the kernel that
        # produces it (upsample backward, stride-2 conv data gradient), autograd never sums
        p5, l5 = F_.fan2(p5)
        p4, l4 = F_.fan2(joined(self.h1, (lambda o: F_.Upsample2x.apply(p5, o, l5), p5.shape[1]), p4))
        p3, l3 = F_.fan2(joined(self.h2, (lambda o: F_.Upsample2x.apply(p4, o, l4), p4.shape[1]), p3))
        p4b, l4b = F_.fan2(joined(self.h4, (lambda o: self.h3(p3, out=o, res_link=l3), self.h3.conv.out_channels), F_.stash(p4, l4)))
        p5b = joined(self.h6, (lambda o: self.h5(p4b, out=o, res_link=l4b), self.h5.conv.out_channels), F_.stash(p5, l5))
        return F_.stash(p3, l3), F_.stash(p4b, l4b), p5b
