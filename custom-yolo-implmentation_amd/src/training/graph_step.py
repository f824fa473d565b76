"""Launch-bound inner loop as HIP graphs, with the data-parallel gradient exchange overlapped with backward.

One training step of the small model is ~700 short kernels; issued eagerly from Python the GPU idles between them.
`TrainStepRunner` captures the step on static buffers and replays it.

Single process: ONE graph = forward + loss + backward + optimizer.

With a process group (DDP semantics: every rank ends the step with the mean gradient, src/training/utils_train.py:190)
the backward is cut at the backbone / neck boundary into two autograd stages and the gradients into two flat
communication buffers ("buckets", output -> input order like DistributedDataParallel's):

    graph A   forward, loss, backward of head + neck, pack their gradients into bucket A
    all_reduce(bucket A)            async on RCCL's stream ...
    graph B   ... while the backward of the backbone runs; pack its gradients into bucket B
    all_reduce(bucket B)
    graph C   unpack both buckets into the .grad tensors, optimizer step

so only the second, smaller exchange is exposed.  xGMI is point-to-point: two large messages per step instead of
DDP's default 25 MB bucket train; the buckets are bf16-compressed on request.  Pack and unpack are ONE launch each over a
device job table (csrc/multi_copy.hip), the collectives are the only host-issued work between three graph replays.
"""
import gc
import os

import torch
import torch.distributed as dist

from src.hipops import functions as F_
from src.hipops import lib
from src.hipops import ops


class GradBuckets:
    """Flat communication buffers for the parameter gradients of the backward stages, and the one-launch pack / unpack
    between them and the .grad tensors.  The job tables hold raw pointers: they are (re)built when the gradient tensors
    change (every eager step; once per capture, where the upload is deferred until the capture has ended)."""

    ALIGN = 16          # elements: every tensor's slice starts 32-byte aligned (vector path of k_multi_copy)

    def __init__(self, stage_params, comm_dtype):
        self.stage_params = [[p for p in ps if p.requires_grad] for ps in stage_params]
        self.comm_dtype = comm_dtype
        self.flats = None
        self.plans = {}                     # (stage, direction) -> dict(host, dev, njobs, nchunks, ptrs)
        self._pending = []

    def _ensure_flats(self):
        if self.flats is not None:
            return
        self.flats, self.slices = [], []
        for ps in self.stage_params:
            off, sl = 0, []
            for p in ps:
                sl.append((off, p.numel()))
                off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            dtype = self.comm_dtype or (ps[0].dtype if ps else torch.float32)
            dev = ps[0].device if ps else "cpu"
            # zero once: the alignment gaps travel through the all-reduce too and must stay finite
            self.flats.append(torch.zeros(max(off, 1), dtype=dtype, device=dev))
            self.slices.append(sl)

    def _plan(self, stage, unpack):
        """Job table of one stage and direction for the CURRENT gradient tensors."""
        self._ensure_flats()
        ps = [p for p in self.stage_params[stage] if p.grad is not None]
        flat = self.flats[stage]
        key = (stage, unpack)
        plan = self.plans.get(key)
        ptrs = tuple(p.grad.data_ptr() for p in ps)
        if plan is not None and plan["ptrs"] == ptrs:
            return plan
        capturing = flat.is_cuda and torch.cuda.is_current_stream_capturing()
        jb = lib.query("yolo_copy_job_bytes")
        n = len(ps)
        if plan is None or plan["host"].numel() != n * jb:
            if capturing:
                raise RuntimeError("GradBuckets: run one eager step before capturing (the job tables are allocated there)")
            host = torch.zeros(max(n, 1) * jb, dtype=torch.uint8)
            plan = self.plans[key] = dict(host=host.pin_memory() if flat.is_cuda else host,
                                          dev=torch.empty(max(n, 1) * jb, dtype=torch.uint8, device=flat.device))
        by_param = {id(p): sl for p, sl in zip(self.stage_params[stage], self.slices[stage])}
        grads, views = [], []
        for i, p in enumerate(ps):
            off, numel = by_param[id(p)]
            g = p.grad
            if not g.is_contiguous():
                raise RuntimeError("GradBuckets needs contiguous gradients")
            v = flat[off:off + numel]
            grads.append(g.view(-1))
            views.append(v)
            src, dst = (v, g) if unpack else (g, v)
            lib.call("yolo_copy_job_fill", plan["host"].data_ptr(), i, src.data_ptr(), ops.dt(src), dst.data_ptr(), ops.dt(dst), numel)
        plan["nchunks"] = lib.query("yolo_copy_jobs_finalize", plan["host"].data_ptr(), n) if n else 0
        plan["njobs"], plan["ptrs"] = n, ptrs
        plan["srcs"], plan["dsts"] = (views, grads) if unpack else (grads, views)
        if capturing:
            self._pending.append(plan)      # the upload must not be captured: finish_capture() does it
        else:
            plan["dev"].copy_(plan["host"], non_blocking=False)
        return plan

    def pack(self, stage):
        ops.bucket_copy(self._plan(stage, False))

    def unpack(self, stage, scale=1.0):
        ops.bucket_copy(self._plan(stage, True), scale)

    def finish_capture(self):
        for plan in self._pending:
            plan["dev"].copy_(plan["host"])
        self._pending = []
        self.captured = {k: (v["host"].clone(), dict(v)) for k, v in self.plans.items()}

    def restore_capture(self):
        """After an eager step between replays rebuilt the tables for its own gradient tensors: put the captured ones back."""
        for k, (host, saved) in getattr(self, "captured", {}).items():
            plan = self.plans[k]
            if plan["ptrs"] != saved["ptrs"]:
                plan["host"].copy_(host)
                plan["dev"].copy_(plan["host"])
                plan.update({f: saved[f] for f in ("ptrs", "njobs", "nchunks", "srcs", "dsts")})


class TrainStepRunner:
    def __init__(self, model, criterion, optimizer, precision="bfloat16", use_graph=True, grad_comm_dtype=None,
                 force_comm=False, buckets=2):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.amp_dtype = {"bfloat16": torch.bfloat16, "float16": torch.float16}.get(precision)
        self.use_graph = use_graph
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.comm_dtype = grad_comm_dtype
        # the collective path; force_comm runs it on a one-rank group too (single-GPU test of the multi-graph step)
        self.comm = self.world > 1 or (force_comm and dist.is_available() and dist.is_initialized())
        self.params = [p for p in model.parameters() if p.requires_grad]
        if hasattr(model, "_prepack"):          # plain local parameters here: pack all conv weights in one launch
            model.prepack = True
        # backward stages, output -> input: [head + neck, backbone] when the model exposes the cut (Model.stage_cut)
        # and two buckets are asked for, else one stage
        self.staged = self.comm and buckets >= 2 and hasattr(model, "stage_cut") and hasattr(model, "net")
        if self.staged:
            back = {id(p) for p in model.net.parameters()}
            stages = [[p for p in self.params if id(p) not in back], [p for p in self.params if id(p) in back]]
        else:
            stages = [self.params]
        self.buckets = GradBuckets(stages, grad_comm_dtype) if self.comm else None
        self.avg_in_collective = self.comm and dist.get_backend() == "nccl"       # gloo has no AVG
        # fp16: dynamic loss scaling kept on the device (no GradScaler round trips: the step stays capturable)
        self.amp = None
        if precision == "float16":
            from src.training.fused_adamw import DeviceGradScaler, HipAdamW
            if not isinstance(optimizer, HipAdamW) or len(optimizer.param_groups) != 1:
                raise RuntimeError("TrainStepRunner(precision='float16') needs HipAdamW with one parameter group: the loss "
                                   "scale, the overflow check and the skipped / unscaled update run on the device")
            self.amp = getattr(optimizer, "device_amp", None) or DeviceGradScaler(self.params[0].device)
            optimizer.device_amp = self.amp
        self.graph = None
        self.graph_b = None                     # staged: backward of the backbone + its pack
        self.graph2 = None                      # comm: unpack + optimizer, replayed after the all-reduces
        self.opt_in_graph = False
        self.static = None
        self.loss = None
        self.scalars = None
        self._cut = None

    # -------------------------------------------------------------------------------------------
    def _lazy(self, device):
        # parameter-gradient work is queued for the side stream and joined lazily, several layers per cross-stream
        # sync point (functions._wgrad_overlapped); nothing reads a gradient before the join at the end of the stage.
        # Only with empty .grad fields: autograd would ADD a new gradient to an existing one right away, on this
        # stream, before the side stream has produced it.  YOLO_LAZY_JOIN=0 keeps the per-layer fork/join (A/B runs)
        return device.type == "cuda" and os.environ.get("YOLO_LAZY_JOIN", "1") == "1" and all(p.grad is None for p in self.params)

    def _stage_a(self, images, packed):
        """forward + loss + backward down to the stage cut (the whole backward when there is no cut)."""
        dev = images.device
        cut = {}
        if self.staged:
            def at_cut(feats):
                cut["out"] = list(feats)
                cut["leaf"] = [t.detach().requires_grad_(True) for t in feats]
                return cut["leaf"]
            self.model.stage_cut = at_cut
        F_.LAZY_WGRAD_JOIN = self._lazy(dev)
        F_.LOSS_SCALE = self.amp.scale if self.amp is not None else None
        try:
            with torch.autocast(dev.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                preds, anchors, strides = self.model(images)
                loss, ld = self.criterion(preds, packed, anchors, strides)
            F_.LOSS_SCALE = None
            # the loss tensor is the fused loss node's own output: the seed of backward() (ones) reaches it unchanged
            F_.UNIT_LOSS_SEED = type(loss.grad_fn).__name__ == "DflQflLossBackward" and os.environ.get("YOLO_UNIT_SEED", "1") == "1"
            loss.backward()
        except BaseException:
            # stage B will not run: the lazy-join flag must not leak into a later plain backward (AccumulateGrad would
            # add to / clone a gradient the side stream has not written yet)
            F_.LAZY_WGRAD_JOIN = False
            self._cut = None
            raise
        finally:
            F_.UNIT_LOSS_SEED = False
            F_.LOSS_SCALE = None
            if self.staged:
                self.model.stage_cut = None
            if not self.staged:
                F_.LAZY_WGRAD_JOIN = False
            if dev.type == "cuda":
                F_.join_wgrad_stream(dev)
        self._cut = cut if self.staged else None
        return loss, ld

    def _stage_b(self, device):
        """backward of the backbone from the gradients the first stage left at the cut."""
        if self._cut is None:
            return
        cut, self._cut = self._cut, None
        try:
            pairs = [(o, l.grad) for o, l in zip(cut["out"], cut["leaf"]) if l.grad is not None]
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        finally:
            F_.LAZY_WGRAD_JOIN = False
            if device.type == "cuda":
                F_.join_wgrad_stream(device)

    def _fwd_bwd(self, images, packed):
        loss, ld = self._stage_a(images, packed)
        self._stage_b(images.device)
        ops.ACTIVE_PACK_PLAN = None         # the packed weights are stale once the optimizer has stepped
        return loss, ld

    # ---- the exchange: the same buckets, reduction and dtype whether the step is replayed or eager
    def _all_reduce(self, stage, async_op=False):
        flat = self.buckets.flats[stage]
        if self.avg_in_collective:
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op)
        return dist.all_reduce(flat, async_op=async_op)

    def _unpack_scale(self):
        return 1.0 if self.avg_in_collective else 1.0 / self.world

    def _exchange_eager(self):
        if not self.comm:
            return
        nst = len(self.buckets.stage_params)
        works = []
        for s in range(nst):
            self.buckets.pack(s)
            works.append(self._all_reduce(s, async_op=True))
        for w in works:
            w.wait()
        for s in range(nst):
            self.buckets.unpack(s, self._unpack_scale())

    def _eager_step(self, images, packed):
        self.optimizer.zero_grad(set_to_none=True)
        loss, ld = self._fwd_bwd(images, packed)
        self._exchange_eager()
        self.optimizer.step()
        return loss, ld

    def _probe_accumulators(self):
        """Hooks on every parameter's AccumulateGrad node that note the stream the node runs on (the engine makes a node's
        own stream current around it).  Returns a function: number of parameters whose node ran on another stream than the
        current one at the time of THIS call; the hooks are removed by it."""
        want = torch.cuda.current_stream().cuda_stream
        seen, handles, nodes = [], [], []
        for p in self.params:
            node = p.view_as(p).grad_fn.next_functions[0][0]        # the (cached) AccumulateGrad node of the leaf
            if node is None:
                continue
            nodes.append(node)                                      # alive until the step has run: the step uses THESE nodes
            handles.append(node.register_hook(lambda *_: seen.append(torch.cuda.current_stream().cuda_stream)))

        def result():
            for h in handles:
                h.remove()
            nodes.clear()
            return sum(1 for st in seen if st != want)
        return result

    # -------------------------------------------------------------------------------------------
    def capture(self, images, packed, warmup=3):
        """Warm up on a side stream (allocator + lazy inits), then capture on static inputs."""
        self.static = (images, packed)
        self.capture_refused = None
        foreign = 0
        if not self.use_graph:
            return self
        side = self.stream = F_.step_stream(images.device)      # warm-up and capture on the stream DDP was built on
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(max(warmup, 1)):
                probe = self._probe_accumulators() if i == 0 else None
                _, ld = self._eager_step(images, packed)
                self.warm_scalars = ld._scalars         # the loss scalars of the last warm-up step (a real step)
                if probe is not None:
                    foreign = probe()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if foreign:
            # autograd runs a parameter's AccumulateGrad node on the stream the node was CREATED on.  Somebody keeps nodes
            # alive that were created elsewhere (a DistributedDataParallel built on the default stream does): they would
            # drag that stream into the capture and hipStreamEndCapture dies (tools/capture_probe.py).  Measured on this
            # very step, not assumed from a flag: stay on the eager step.
            import warnings
            warnings.warn(f"TrainStepRunner: {foreign} parameter gradients are accumulated on a stream other than the step "
                          "stream (AccumulateGrad nodes created there are kept alive, e.g. by a DistributedDataParallel built "
                          "on the default stream): the step is NOT captured, eager steps instead", stacklevel=2)
            self.capture_refused = "accumulate-grad nodes on a foreign stream"
            return self
        capturable = all(g.get("capturable", False) for g in self.optimizer.param_groups)
        want_opt = not self.comm and capturable
        self.optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        # No garbage collection while capturing: a collected object that owns pinned host memory (an optimizer's job
        # table, StaticTargets of an earlier runner) is freed through the caching host allocator, which records and
        # queries events -- not allowed on the capturing thread; the process then aborts, depending on when the
        # collector happens to run (seen as a run-order dependent abort while packing gradients).
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
        dev = images.device
        with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
            self.loss, ld = self._stage_a(images, packed)
            self.scalars = ld._scalars
            if not self.staged:
                self._stage_b(dev)
                ops.ACTIVE_PACK_PLAN = None
            if want_opt:
                self.optimizer.step()
            elif self.comm:
                self.buckets.pack(0)
        self.graph, self.opt_in_graph = g, want_opt
        if self.staged:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=g.pool(), stream=self.stream, capture_error_mode="thread_local"):
                self._stage_b(dev)
                ops.ACTIVE_PACK_PLAN = None
                self.buckets.pack(1)
            self.graph_b = gb
        finish = getattr(self.optimizer, "finish_capture", None)     # HipAdamW: upload the job table recorded in capture
        if self.comm and capturable:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool(), stream=self.stream, capture_error_mode="thread_local"):
                for s in range(len(self.buckets.stage_params)):
                    self.buckets.unpack(s, self._unpack_scale())
                self.optimizer.step()
            self.graph2 = g2
        if self.comm:
            self.buckets.finish_capture()
        if finish is not None:
            finish()
        return self

    def capture_for_batches(self, images, gt_boxes_list, boxes_per_image=128, warmup=3):
        """Capture on buffers that `step_batch` refills: a static image tensor and fixed-capacity target buffers
        (`StaticTargets`, capacity = boxes_per_image x batch size).  The data loader must keep batch size and
        resolution fixed (the reference's does: drop_last).  Returns None when the batch does not fit the capacity
        (the caller steps it eagerly and captures on a later batch)."""
        from src.model.losses import StaticTargets
        self.static_images = images.detach().clone()
        self.static_targets = StaticTargets(len(gt_boxes_list), boxes_per_image * len(gt_boxes_list), images.device)
        if not self.static_targets.load(gt_boxes_list):
            self.static_images = self.static_targets = None
            return None
        return self.capture(self.static_images, self.static_targets, warmup=warmup)

    def fits(self, images, gt_boxes_list):
        """Whether `step_batch` can take this batch (host-side check, no device work)."""
        st = getattr(self, "static_images", None)
        return st is not None and images.shape == st.shape and images.dtype == st.dtype and \
            self.static_targets.fits(gt_boxes_list)

    def step_batch(self, images, gt_boxes_list):
        """One optimizer step on a NEW batch of the captured shape: refill the static buffers, replay.  Returns the
        device loss tensor, or None if the batch does not fit (other batch size / resolution, more boxes than the
        capacity): the caller then runs that batch eagerly."""
        if not self.fits(images, gt_boxes_list) or not self.static_targets.load(gt_boxes_list):
            return None
        self.static_images.copy_(images, non_blocking=True)
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
            works = [self._all_reduce(0, async_op=True)]      # runs on the collective's stream beside graph B
            if self.graph_b is not None:
                self.graph_b.replay()
                works.append(self._all_reduce(1, async_op=True))
            for w in works:
                w.wait()
            if self.graph2 is not None:
                self.graph2.replay()
                return self.loss
            for s in range(len(self.buckets.stage_params)):
                self.buckets.unpack(s, self._unpack_scale())
        self.optimizer.step()
        return self.loss
