"""Model = Backbone + Neck + Head with the reference's API (src/model/model_builder.py:13-139):
forward(x) -> (preds[N,64+nc,M], anchors[2,M], strides[1,M]); fuse(); load_weights(); inference()."""
import os
from typing import List

import torch
from torch import nn

from src.hipops import functions as F_
from src.hipops import ops
from src.model.backbone import Backbone
from src.model.head import Head
from src.model.infer_graph import InferenceGraphs
from src.model.model_blocks import Conv
from src.model.neck import Neck
from src.utils.model_utils import fuse_conv, non_max_suppression


NECK_DIRECT = os.environ.get("YOLO_NECK_DIRECT", "1") == "1"


class Model(nn.Module):
    # TrainStepRunner's staged backward: a callable that receives the backbone's outputs and returns what the neck should
    # read instead (detached leaves), cutting the autograd graph at the backbone / neck boundary
    stage_cut = None
    # inference() replays forward + decode as one hipGraph per input shape (src/model/infer_graph.py); False, or
    # YOLO_INFER_GRAPH=0 in the environment, issues the launches one by one as forward() does
    graph_inference = os.environ.get("YOLO_INFER_GRAPH", "1") == "1"

    def __init__(self, width: List[int], depth: List[int], csp: List[bool], num_classes: int):
        super().__init__()
        self.net = Backbone(width, depth, csp)
        self.fpn = Neck(width, depth, csp)
        self.num_classes = num_classes
        self.head = Head(num_classes, (width[3], width[4], width[5]))
        # The reference measures the strides with a 640x640 CPU dummy pass (:37-45); they are fixed by
        # the architecture (three / four / five stride-2 stages), so no pass is needed here.
        self.head.stride = torch.tensor([8.0, 16.0, 32.0])
        self.stride = self.head.stride
        for m in self.modules():                 # BatchNorm's num_batches_tracked: one multi-tensor add per
            if type(m) is Conv:                  # forward instead of one tiny kernel per layer
                m._count_batches = False
        self._infer_graphs = None
        self._bn_convs = None

    def _apply(self, fn, *args, **kwargs):       # .to() / .half() / .cuda(): new storages -- captured graphs point at the old
        self._infer_graphs = None
        self._bn_convs = None
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x):
        if not self.training:
            # packed weights served by a training pass's plan predate the last optimizer step (a raw kernel updates the
            # parameters without touching their version counters): evaluation always packs afresh
            ops.ACTIVE_PACK_PLAN = None
            return self.head(list(self.fpn(self.net(x))))
        convs = self._bn_convs                   # (walking the module tree costs ~0.8 ms per eager step on preset s)
        if convs is None:
            convs = self._bn_convs = [m for m in self.modules() if type(m) is Conv and hasattr(m, "norm")]
        torch._foreach_add_([m.norm.num_batches_tracked for m in convs], 1)
        self._prepack(x, convs)
        # one zeroed pool for every layer's BatchNorm accumulators of this pass (a single memset)
        F_.BnArena.current = F_.BnArena(x.device, F_.BnArena.elems_for([m.conv.out_channels for m in convs]))
        try:
            # the neck's concat buffers; the backbone writes p3 / p4 / p5 into them (YOLO_NECK_DIRECT=0: A/B runs)
            bufs, outs = self.fpn.alloc(x) if NECK_DIRECT else (None, None)
            feats = self.net(x, outs)
            if self.stage_cut is not None:
                feats = self.stage_cut(feats)
            return self.head(list(self.fpn(feats, bufs)))
        finally:
            F_.BnArena.current = None

    def _prepack(self, x, convs):
        """Pack every dense conv weight (forward + dgrad forms) with one launch instead of two per layer.
        Only for plain local parameters; sharded/wrapped parameters (FSDP) fall back to per-layer packing."""
        ops.ACTIVE_PACK_PLAN = None
        if getattr(self, "prepack", None) is False:  # (True: TrainStepRunner; unset: decided per pass from the weights)
            return
        dense = [(m.conv.weight, m._k, m._s) for m in convs if not m._dw and m.conv.in_channels != 3]
        dense += [(b[-1].weight, 1, 1) for br in (self.head.box, self.head.cls) for b in br]
        # plain, resident local tensors only -- never what a sharding wrapper manages: FSDP2 holds DTensors until a group is
        # unsharded, FSDP1's original parameters are views of a flat parameter whose storage is freed between uses
        for w, _, _ in dense:
            if type(w.data) is not torch.Tensor or not w.is_cuda or w.dim() != 4 or getattr(w, "_fsdp_flattened", False) or \
                    w.untyped_storage().size() < w.numel() * w.element_size():
                return
        T = F_.compute_dtype(x, dense[0][0])
        key = (T, tuple(w.data_ptr() for w, _, _ in dense))
        plan = getattr(self, "_pack_plan", None)
        if plan is None or (plan.dtype, plan.key) != key:
            plan = self._pack_plan = ops.WeightPackPlan(dense, T)
        plan.run()
        ops.ACTIVE_PACK_PLAN = plan

    def fuse(self):
        """Fold every Conv's BatchNorm into its conv (inference only), reference :52-58."""
        self._infer_graphs = None
        self._bn_convs = None
        for m in self.modules():
            if type(m) is Conv and hasattr(m, "norm"):
                m.conv = fuse_conv(m.conv, m.norm)
                m.forward = m.fuse_forward
                delattr(m, "norm")
        return self

    def load_weights(self, weights_path):
        checkpoint = torch.load(weights_path, map_location=next(self.parameters()).device)
        state = checkpoint["model_state"] if isinstance(checkpoint, dict) and "model_state" in checkpoint else checkpoint
        # checkpoints written under DDP / FSDP1 carry wrapper prefixes ("module." ...): the reference's own load fails on
        # them (notebooks/04); keys are matched by their canonical names here
        from src.training.utils_train import canonical_state_dict
        self.load_state_dict(canonical_state_dict(state))
        self._infer_graphs = None
        print(f"Weights loaded successfully from {weights_path}")

    def inference(self, image, conf_thres=0.25, iou_thres=0.45):
        """Detections [x1,y1,x2,y2,conf,cls] per image; class scores are the raw logits, as in the
        reference (:115-139).  Decode (DFL expectation -> xywh -> *stride) is one fused kernel."""
        graphs = None
        if self.graph_inference:
            if self._infer_graphs is None:
                self._infer_graphs = InferenceGraphs()
            graphs = self._infer_graphs
        if graphs is None or not graphs.all_eval(self):
            self.eval()
        if isinstance(image, str):
            from PIL import Image
            image = Image.open(image).convert("RGB")
        if not isinstance(image, torch.Tensor):
            try:
                from PIL import Image
                is_pil = isinstance(image, Image.Image)
            except ImportError:
                is_pil = False
            if not is_pil:
                raise ValueError("Unsupported image type. Must be path, PIL Image, or Tensor.")
            from src.data.transforms import get_val_transforms
            image = get_val_transforms(device=next(self.parameters()).device)([image])[0]      # resize + normalise on the device
        if image.dim() == 3:
            image = image.unsqueeze(0)
        image = image.to(next(self.parameters()).device)
        with torch.no_grad():
            if graphs is not None and image.is_cuda:
                dets = graphs.run(self, image, conf_thres, iou_thres)
                if dets is not None:
                    return dets
            preds, anchors, strides = self.forward(image)
            y = ops.head_decode(preds, anchors, strides, self.head.nc)
            return non_max_suppression(y, conf_thres=conf_thres, iou_thres=iou_thres, nc=self.num_classes)
