"""Detection head: per level a box branch (3x3, 3x3, 1x1 -> 4*16 DFL logits) and a class branch
(dw3x3, 1x1, dw3x3, 1x1, 1x1 -> nc logits); outputs are written transposed straight into the
(N, 64+nc, M) prediction tensor.  Mirrors the reference's src/model/head.py:35-121."""
import math
from typing import List

import torch
from torch import nn

from src.hipops import functions as F_
from src.model.model_blocks import DFL, Conv
from src.utils.model_utils import make_anchors_cached


class Head(nn.Module):
    anchors = torch.empty(0)
    strides = torch.empty(0)

    def __init__(self, nc: int = 1, filters: List[int] = []):
        super().__init__()
        self.ch = 16
        self.nc = nc
        self.nl = len(filters)
        self.no = nc + self.ch * 4
        self.stride = torch.zeros(self.nl)
        box = max(64, filters[0] // 4)
        cls = max(80, filters[0], self.nc)
        self.dfl = DFL(self.ch)
        self.box = nn.ModuleList(
            nn.Sequential(Conv(x, box, nn.SiLU(), k=3, p=1), Conv(box, box, nn.SiLU(), k=3, p=1),
                          nn.Conv2d(box, out_channels=4 * self.ch, kernel_size=1)) for x in filters)
        self.cls = nn.ModuleList(
            nn.Sequential(Conv(x, x, nn.SiLU(), k=3, p=1, g=x), Conv(x, cls, nn.SiLU()),
                          Conv(cls, cls, nn.SiLU(), k=3, p=1, g=cls), Conv(cls, cls, nn.SiLU()),
                          nn.Conv2d(cls, out_channels=self.nc, kernel_size=1)) for x in filters)
        self.initialize_biases()

    def initialize_biases(self):
        """Class-logit bias = log(p/(1-p)) with p = 0.01 (reference :66-74)."""
        b = math.log(1e-2 / (1 - 1e-2))
        for branch in self.cls:
            nn.init.constant_(branch[-1].bias, b)

    @staticmethod
    def _branch(seq, x, link=None):
        x = seq[0](x, res_link=link)
        for m in seq[1:-1]:
            x = m(x)
        last = seq[-1]
        return F_.ConvBias.apply(x, last.weight, last.bias, 1, 1)

    def forward(self, x):
        # a level feeds its box and its class branch: the second branch's data gradient is added to the first's (F_.fan2)
        pairs = [F_.fan2(t) for t in x]
        xb = xc = [p[0] for p in pairs]
        links = [p[1] for p in pairs]
        hooked = any(m._forward_hooks or m._forward_pre_hooks for m in self.cls.modules())    # a user hook would read a
        if self.training and x[0].is_cuda and F_.HEAD_TWO_STREAMS and torch.is_grad_enabled() and not hooked:    # side-stream tensor unsynchronised
            # forward of the class branches on the auxiliary stream, beside the box branches (one fork, one join).
            # Only the LAUNCHES move (F_.FWD_STREAM, inside each node's forward): autograd still records the nodes
            # on the current stream, so the backward stays single-stream -- backward nodes that autograd itself runs
            # on a second stream crash graph capture on this stack
            dev = x[0].device
            cur, side = torch.cuda.current_stream(dev), F_.side_stream(dev)
            side.wait_stream(cur)
            F_.FWD_STREAM = side
            try:
                cls_outs = [self._branch(self.cls[i], xc[i], links[i]) for i in range(self.nl)]
            finally:
                F_.FWD_STREAM = None
            box_outs = [self._branch(self.box[i], xb[i], links[i]) for i in range(self.nl)]
            cur.wait_stream(side)
            outs = [o for pair in zip(box_outs, cls_outs) for o in pair]
        else:
            outs = []
            for i in range(self.nl):
                outs += [self._branch(self.box[i], xb[i], links[i]), self._branch(self.cls[i], xc[i], links[i])]
        preds = F_.HeadPack.apply(*outs)
        shapes = tuple((int(o.shape[2]), int(o.shape[3])) for o in outs[::2])
        anchors, strides = make_anchors_cached(shapes, tuple(float(s) for s in self.stride), preds.dtype, preds.device)
        return preds, anchors, strides
