"""Launch-bound inner loop as a HIP graph.

One training step of the small model is ~1000 short kernels; issued eagerly from Python the GPU idles
between them.  `TrainStepRunner` captures forward + loss + backward (+ the optimizer step when it is
capturable) into one hipGraph on static buffers and replays it; data-parallel gradient averaging
(DDP semantics: mean over ranks, src/training/utils_train.py:190) runs between the backward graph and
the optimizer as ONE flat RCCL all-reduce -- xGMI is point-to-point, one large message beats DDP's default
25 MB bucket train for a 38 MB model.  With a process group the step is TWO graphs around that collective:
graph 1 = forward + loss + backward + "pack all gradients into the flat communication buffer" (one multi-tensor
copy, bf16-compressed if asked), then `all_reduce(AVG)` on the flat buffer, then graph 2 = "unpack into the
gradients" + the (capturable) optimizer step -- a dozen launches per step from the host instead of ~300.
"""
import gc
import os

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors

from src.hipops import functions as F_


class TrainStepRunner:
    def __init__(self, model, criterion, optimizer, precision="bfloat16", use_graph=True, grad_comm_dtype=None,
                 force_comm=False):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.amp_dtype = {"bfloat16": torch.bfloat16, "float16": torch.float16}.get(precision)
        self.use_graph = use_graph
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.comm_dtype = grad_comm_dtype
        # the collective path; force_comm runs it on a one-rank group too (single-GPU test of the two-graph step)
        self.comm = self.world > 1 or (force_comm and dist.is_available() and dist.is_initialized())
        self.params = [p for p in model.parameters() if p.requires_grad]
        if hasattr(model, "_prepack"):          # plain local parameters here: pack all conv weights in one launch
            model.prepack = True
        self.graph = None
        self.graph2 = None                      # world > 1: unpack + optimizer, replayed after the all-reduce
        self.flat = None
        self.opt_in_graph = False
        self.static = None
        self.loss = None
        self.scalars = None

    # -------------------------------------------------------------------------------------------
    def _fwd_bwd(self, images, packed):
        dev_type = images.device.type
        # parameter-gradient work is queued for the side stream and joined lazily, several layers per cross-stream
        # sync point (functions._wgrad_overlapped); nothing reads a gradient before this method returns.
        # YOLO_LAZY_JOIN=0 keeps the per-layer fork/join (diagnosis / A-B runs)
        # Only with empty .grad fields: autograd would ADD a new gradient to an existing one right away, on this
        # stream, before the side stream has produced it.
        F_.LAZY_WGRAD_JOIN = dev_type == "cuda" and os.environ.get("YOLO_LAZY_JOIN", "1") == "1" and \
            all(p.grad is None for p in self.params)
        try:
            with torch.autocast(dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                preds, anchors, strides = self.model(images)
                loss, ld = self.criterion(preds, packed, anchors, strides)
            loss.backward()
        finally:
            F_.LAZY_WGRAD_JOIN = False
            if dev_type == "cuda":
                F_.join_wgrad_stream(images.device)
        return loss, ld

    def _allreduce(self):
        if not self.comm:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        flat = _flatten_dense_tensors(grads)
        if self.comm_dtype is not None and flat.dtype != self.comm_dtype:
            comp = flat.to(self.comm_dtype)
            dist.all_reduce(comp)
            flat = comp.to(flat.dtype)
        else:
            dist.all_reduce(flat)
        flat.div_(self.world)
        for g, f in zip(grads, _unflatten_dense_tensors(flat, grads)):
            g.copy_(f)

    def _eager_step(self, images, packed):
        self.optimizer.zero_grad(set_to_none=True)
        loss, ld = self._fwd_bwd(images, packed)
        self._allreduce()
        self.optimizer.step()
        return loss, ld

    # -------------------------------------------------------------------------------------------
    def capture(self, images, packed, warmup=3):
        """Warm up on a side stream (allocator + lazy inits), then capture on static inputs."""
        self.static = (images, packed)
        if not self.use_graph:
            return self
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                _, ld = self._eager_step(images, packed)
                self.warm_scalars = ld._scalars         # the loss scalars of the last warm-up step (a real step)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        capturable = all(g.get("capturable", False) for g in self.optimizer.param_groups)
        want_opt = not self.comm and capturable
        self.optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        # No garbage collection while capturing: a collected object that owns pinned host memory (an optimizer's job
        # table, StaticTargets of an earlier runner) is freed through the caching host allocator, which records and
        # queries events -- not allowed on the capturing thread; the process then aborts, depending on when the
        # collector happens to run (seen as a run-order dependent abort in _pack_grads).
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._capture_graphs(g, images, packed, want_opt, capturable)
        finally:
            if gc_was_on:
                gc.enable()

    def _capture_graphs(self, g, images, packed, want_opt, capturable):
        # thread_local: RCCL's watchdog thread polls events of earlier collectives; in the default (global) mode that
        # query is an error while ANY thread captures
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.loss, ld = self._fwd_bwd(images, packed)
            self.scalars = ld._scalars
            if want_opt:
                self.optimizer.step()
            elif self.comm:
                self._pack_grads()
        self.graph, self.opt_in_graph = g, want_opt
        finish = getattr(self.optimizer, "finish_capture", None)     # HipAdamW: upload the job table recorded in capture
        if self.comm and capturable:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool(), capture_error_mode="thread_local"):
                self._unpack_grads()
                self.optimizer.step()
            self.graph2 = g2
        if finish is not None:
            finish()
        return self

    # ---- flat communication buffer (world > 1, graph mode): gradients are static tensors after capture
    def _pack_grads(self):
        grads = [p.grad for p in self.params if p.grad is not None]
        if self.flat is None:
            dt = self.comm_dtype or grads[0].dtype
            self.flat = torch.empty(sum(g.numel() for g in grads), dtype=dt, device=grads[0].device)
            self._views, off = [], 0
            for g in grads:
                self._views.append(self.flat[off:off + g.numel()].view(g.shape))
                off += g.numel()
        self._grads = grads
        torch._foreach_copy_(self._views, grads)             # one multi-tensor launch per ~100 tensors, casts included

    def _unpack_grads(self):
        torch._foreach_copy_(self._grads, self._views)

    def _reduce_flat(self):
        if dist.get_backend() == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
        else:                                                 # gloo has no AVG
            dist.all_reduce(self.flat)
            self.flat.div_(self.world)

    def capture_for_batches(self, images, gt_boxes_list, boxes_per_image=128, warmup=3):
        """Capture on buffers that `step_batch` refills: a static image tensor and fixed-capacity target buffers
        (`StaticTargets`, capacity = boxes_per_image x batch size).  The data loader must keep batch size and
        resolution fixed (the reference's does: drop_last)."""
        from src.model.losses import StaticTargets
        self.static_images = images.detach().clone()
        self.static_targets = StaticTargets(len(gt_boxes_list), boxes_per_image * len(gt_boxes_list), images.device)
        if not self.static_targets.load(gt_boxes_list):
            raise RuntimeError("capture_for_batches: the first batch does not fit the target capacity")
        return self.capture(self.static_images, self.static_targets, warmup=warmup)

    def step_batch(self, images, gt_boxes_list):
        """One optimizer step on a NEW batch of the captured shape: refill the static buffers, replay.  Returns the
        device loss tensor, or None if the batch does not fit (other batch size / resolution, more boxes than the
        capacity): the caller then runs that batch eagerly."""
        st = getattr(self, "static_images", None)
        if st is None or images.shape != st.shape or images.dtype != st.dtype:
            return None
        if not self.static_targets.load(gt_boxes_list):
            return None
        st.copy_(images, non_blocking=True)
        return self.step()

    def step(self):
        """One optimizer step on the static batch; returns the (device) loss tensor of that step."""
        images, packed = self.static
        if self.graph is None:
            loss, ld = self._eager_step(images, packed)
            self.scalars = ld._scalars
            return loss
        sync = getattr(self.optimizer, "sync_hyper", None)    # HipAdamW: a scheduler may have changed lr on the host
        if sync is not None:
            sync()
        self.graph.replay()
        if self.opt_in_graph:
            return self.loss
        if self.comm:
            self._reduce_flat()
            if self.graph2 is not None:
                self.graph2.replay()
                return self.loss
            self._unpack_grads()
        self.optimizer.step()
        return self.loss
