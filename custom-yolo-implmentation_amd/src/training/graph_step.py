"""Launch-bound inner loop as a HIP graph.

One training step of the small model is ~1000 short kernels; issued eagerly from Python the GPU idles
between them.  `TrainStepRunner` captures forward + loss + backward (+ the optimizer step when it is
capturable) into one hipGraph on static buffers and replays it; data-parallel gradient averaging
(DDP semantics: mean over ranks, src/training/utils_train.py:190) runs between the backward graph and
the optimizer as ONE flat RCCL all-reduce per dtype bucket -- xGMI is point-to-point, a few large
messages beat DDP's default 25 MB bucket train for a 38 MB model.
"""
import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


class TrainStepRunner:
    def __init__(self, model, criterion, optimizer, precision="bfloat16", use_graph=True, grad_comm_dtype=None):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.amp_dtype = {"bfloat16": torch.bfloat16, "float16": torch.float16}.get(precision)
        self.use_graph = use_graph
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.comm_dtype = grad_comm_dtype
        self.params = [p for p in model.parameters() if p.requires_grad]
        if hasattr(model, "_prepack"):          # plain local parameters here: pack all conv weights in one launch
            model.prepack = True
        self.graph = None
        self.opt_in_graph = False
        self.static = None
        self.loss = None
        self.scalars = None

    # -------------------------------------------------------------------------------------------
    def _fwd_bwd(self, images, packed):
        dev_type = images.device.type
        with torch.autocast(dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            preds, anchors, strides = self.model(images)
            loss, ld = self.criterion(preds, packed, anchors, strides)
        loss.backward()
        return loss, ld

    def _allreduce(self):
        if self.world == 1:
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
                self._eager_step(images, packed)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        want_opt = self.world == 1 and all(g.get("capturable", False) for g in self.optimizer.param_groups)
        self.optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss, ld = self._fwd_bwd(images, packed)
            self.scalars = ld._scalars
            if want_opt:
                self.optimizer.step()
        self.graph, self.opt_in_graph = g, want_opt
        return self

    def step(self):
        """One optimizer step on the static batch; returns the (device) loss tensor of that step."""
        images, packed = self.static
        if self.graph is None:
            loss, ld = self._eager_step(images, packed)
            self.scalars = ld._scalars
            return loss
        self.graph.replay()
        if not self.opt_in_graph:
            self._allreduce()
            self.optimizer.step()
        return self.loss
