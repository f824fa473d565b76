"""BASELINE config 4 ("FSDP full-shard") as a captured step: parameters sharded at rest, gathered for compute.

The reference's `--mode fsdp2` (src/training/utils_train.py:116-165) shards every parameter over the ranks, all-gathers a
group's parameters in the low-precision dtype right before its forward / backward, frees them after, and reduce-scatters
the gradients; the optimizer steps each rank's fp32 shard.  torch's FSDP2 over the HIP-backed model reproduces that
(`prepare_fsdp2_model`, verified in tests/test_gpu_fsdp.py) but is driven by host hooks per parameter group and per
parameter: the step cannot be captured and preset l runs at a quarter of its unsharded rate.

`ShardedStepRunner` keeps the CONTRACT and changes the schedule for a 288 GB part, where re-sharding a 50 MB model inside a
step buys nothing:

  at rest     every rank owns 1/world of ONE flat fp32 master vector + its AdamW moments (ZeRO / FSDP sharding of the
              optimizer state and the master weights)
  compute     the model's parameters are views of one flat low-precision vector (bf16 parameters AND BatchNorm buffers,
              inputs cast, no autocast: the reference's FSDP2 numerics, :146-153; train_model.py:240-245)
  per step    graph A   forward + loss + backward + ONE launch packing every gradient into the flat gradient vector
              reduce_scatter(AVG, low precision)                    each rank gets the mean gradient of ITS shard
              graph C   one-launch AdamW on the fp32 shard + one-launch cast into the rank's slice of the parameter vector
              all_gather(low precision)                             everybody's parameters for the next step

Two collectives of (model size x 2 bytes) per step over RCCL / xGMI instead of ~22 host-scheduled ones, nothing between
them but three graph replays.  fp32 precision: same schedule with fp32 vectors (the master shard IS the rank's slice).
"""
import torch
import torch.distributed as dist

from src.hipops import functions as F_
from src.hipops import lib
from src.hipops import ops
from src.training.graph_step import GradBuckets, TrainStepRunner

_LOWP = {"bfloat16": torch.bfloat16, "float16": torch.float16}


def _copy_plan(src, dst):
    """Job table of ONE k_multi_copy launch dst <- src (dtype cast on the way)."""
    jb = lib.query("yolo_copy_job_bytes")
    host = torch.zeros(jb, dtype=torch.uint8)
    lib.call("yolo_copy_job_fill", host.data_ptr(), 0, src.data_ptr(), ops.dt(src), dst.data_ptr(), ops.dt(dst), src.numel())
    nchunks = lib.query("yolo_copy_jobs_finalize", host.data_ptr(), 1)
    return dict(dev=host.to(dst.device), njobs=1, nchunks=nchunks, srcs=[src], dsts=[dst])


class ShardState:
    """The flat vectors of a model sharded over the current process group (see the module docstring): built once, by
    `ShardedStepRunner` or ahead of it by `get_optimizer` (src/training/utils_train.py) when the optimizer must exist before
    the runner does.  The model's parameters become views of `flat_p`; `master` is this rank's fp32 shard (an nn.Parameter:
    what the optimizer steps)."""

    ALIGN = GradBuckets.ALIGN

    def __init__(self, model, precision):
        if precision == "float16":
            raise RuntimeError("native sharding: float16 needs loss scaling across shards; use bfloat16 or float32")
        self.precision = precision
        self.lowp = _LOWP.get(precision)
        self.group = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.group else 1
        self.rank = dist.get_rank() if self.group else 0
        dev = next(model.parameters()).device
        T = self.lowp or torch.float32
        trainable = [p for p in model.parameters() if p.requires_grad]
        slices, off = [], 0
        for p in trainable:
            slices.append((off, p.numel()))
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        unit = self.world * self.ALIGN
        total = (off + unit - 1) // unit * unit
        self.shard_elems = total // self.world
        full32 = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, (o, n) in zip(trainable, slices):
            full32[o:o + n].copy_(p.detach().reshape(-1).float())
        self.flat_p = full32.to(T) if self.lowp is not None else full32            # what the model computes with
        self.flat_g = torch.zeros(total, dtype=T, device=dev)                       # gaps stay zero
        lo, hi = self.rank * self.shard_elems, (self.rank + 1) * self.shard_elems
        self.p_shard, self.g_shard = self.flat_p[lo:hi], torch.zeros(self.shard_elems, dtype=T, device=dev)
        # fp32 master of this rank's shard: a separate vector under mixed precision, the slice itself in fp32
        master = full32[lo:hi].clone() if self.lowp is not None else self.p_shard
        self.master = torch.nn.Parameter(master, requires_grad=True)
        # ---- the numeric contract: parameters and BatchNorm buffers in the low-precision dtype (reference :146-153)
        with torch.no_grad():
            for p, (o, n) in zip(trainable, slices):
                p.data = self.flat_p[o:o + n].view(p.shape)
            for p in model.parameters():
                if not p.requires_grad and p.is_floating_point() and self.lowp is not None:
                    p.data = p.data.to(self.lowp)                                   # frozen (DFL) weights: cast, not sharded
            if self.lowp is not None:
                for b in model.buffers():
                    b.data = b.data.to(self.lowp)            # every buffer, like the reference (:150-153); counters included
        self.model, self.trainable, self.slices, self.total = model, trainable, slices, total

    def gather_flat(self, shard):
        """The full flat vector (fp32) from every rank's fp32 shard of it -- a collective."""
        shard = shard.detach().float().contiguous()
        if self.group and self.world > 1:
            if dist.get_backend() != "nccl":                 # gloo: sum of zero-padded slices
                full = torch.zeros(self.total, dtype=torch.float32, device=shard.device)
                lo = self.rank * self.shard_elems
                full[lo:lo + self.shard_elems].copy_(shard)
                dist.all_reduce(full)
                return full
            full = torch.empty(self.total, dtype=torch.float32, device=shard.device)
            dist.all_gather_into_tensor(full, shard)
            return full
        return shard

    def full_state_dict(self):
        """{canonical name: FULL fp32 tensor} of the model -- the fp32 masters gathered from every rank (a collective) for
        the trainable parameters, the model's own tensors for buffers and frozen weights: loads into a bare `Model`."""
        full = self.gather_flat(self.master.data)
        by_id = {id(p): (o, n) for p, (o, n) in zip(self.trainable, self.slices)}
        out = {}
        for k, v in self.model.state_dict().items():
            out[k] = v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu()
        for k, p in self.model.named_parameters():
            if id(p) in by_id:
                o, n = by_id[id(p)]
                out[k] = full[o:o + n].view(p.shape).cpu().clone()
        return out

    def full_optimizer_state_dict(self, optimizer):
        """The sharded AdamW state as the state dict of a torch.optim.AdamW over the bare model's trainable parameters (in
        `model.parameters()` order): moments gathered from every rank (a collective) and cut per parameter."""
        st = optimizer.state.get(self.master, {})
        sd = optimizer.state_dict()
        group = dict(sd["param_groups"][0])
        group["params"] = list(range(len(self.trainable)))
        state = {}
        if "exp_avg" in st:
            m1, m2 = self.gather_flat(st["exp_avg"]), self.gather_flat(st["exp_avg_sq"])
            step = torch.as_tensor(st["step"]).detach().float().cpu().reshape(())
            for i, (p, (o, n)) in enumerate(zip(self.trainable, self.slices)):
                state[i] = dict(step=step.clone(), exp_avg=m1[o:o + n].view(p.shape).cpu().clone(),
                                exp_avg_sq=m2[o:o + n].view(p.shape).cpu().clone())
        return dict(state=state, param_groups=[group])


class ShardedStepRunner(TrainStepRunner):
    """See the module docstring.  `optimizer_factory(params) -> optimizer` builds the optimizer over the ONE fp32 shard
    parameter (default: HipAdamW(lr, weight_decay) on the GPU, torch.optim.AdamW elsewhere); or hand in `shard` (a
    ShardState) together with the `optimizer` that was built over `shard.master`."""

    def __init__(self, model, criterion, precision="bfloat16", lr=1e-3, weight_decay=1e-2, optimizer_factory=None, use_graph=True,
                 shard=None, optimizer=None):
        sh = shard if shard is not None else ShardState(model, precision)
        if (shard is None) != (optimizer is None):
            raise ValueError("ShardedStepRunner: pass `shard` and the `optimizer` built over shard.master together")
        self.shard = sh
        for k in ("lowp", "group", "world", "rank", "shard_elems", "flat_p", "flat_g", "p_shard", "g_shard", "master",
                  "trainable", "slices", "total"):
            setattr(self, k, getattr(sh, k))
        dev = self.master.device
        T = self.lowp or torch.float32
        trainable, slices = self.trainable, self.slices
        if optimizer is None:
            if optimizer_factory is None:
                if dev.type == "cuda":
                    from src.training.fused_adamw import HipAdamW
                    optimizer_factory = lambda ps: HipAdamW(ps, lr=lr, weight_decay=weight_decay)
                else:
                    optimizer_factory = lambda ps: torch.optim.AdamW(ps, lr=lr, weight_decay=weight_decay)
            optimizer = optimizer_factory([self.master])
        super().__init__(model, criterion, optimizer, precision="float32", use_graph=use_graph)     # no autocast in FSDP modes
        self.comm, self.staged, self.buckets = False, False, None      # the exchange below replaces the DDP buckets
        # pack: every gradient into its slice of flat_g in one launch (the GradBuckets job table over OUR layout)
        self.pack = GradBuckets([trainable], T)
        self.pack.flats, self.pack.slices = [self.flat_g], [slices]
        self.cast_plan = _copy_plan(self.master.data, self.p_shard) if self.lowp is not None and dev.type == "cuda" else None
        self.graph_c = None
        self.sum_then_scale = self.group and dist.get_backend() != "nccl"          # gloo has no AVG

    # ------------------------------------------------------------------------------------------- the three pieces of a step
    def _reduce_scatter(self):
        if not self.group:
            self.g_shard.copy_(self.flat_g[: self.shard_elems])
            return
        if self.sum_then_scale:
            # gloo (tests, rehearsals on one card): no AVG, no low-precision reduce-scatter -- an fp32 all-reduce, this rank's slice
            g32 = self.flat_g.float()
            dist.all_reduce(g32)
            lo = self.rank * self.shard_elems
            self.g_shard.copy_(g32[lo:lo + self.shard_elems] / self.world)
        else:
            dist.reduce_scatter_tensor(self.g_shard, self.flat_g, op=dist.ReduceOp.AVG)

    def _update_shard(self):
        """AdamW on the fp32 shard from the mean gradient of the shard, then the shard's low-precision image."""
        if self.g_shard.dtype == self.master.dtype:
            self.master.grad = self.g_shard                  # (zero_grad may have dropped it)
        elif self.master.is_cuda:
            self.master.lowp_grad = self.g_shard             # HipAdamW reads a low-precision gradient of an fp32 parameter here
        else:
            self.master.grad = self.g_shard.float()          # torch.optim.AdamW wants the parameter's dtype (CPU tests)
        self.optimizer.step()
        if self.lowp is not None:
            if self.cast_plan is not None:
                ops.bucket_copy(self.cast_plan)
            else:
                self.p_shard.copy_(self.master.data)

    def _all_gather(self):
        if not self.group:
            return
        if self.sum_then_scale:                              # gloo: every rank contributes its slice of a zero vector, summed
            full = torch.zeros(self.total, dtype=torch.float32, device=self.flat_p.device)
            lo = self.rank * self.shard_elems
            full[lo:lo + self.shard_elems].copy_(self.p_shard)
            dist.all_reduce(full)
            self.flat_p.copy_(full)
        else:
            dist.all_gather_into_tensor(self.flat_p, self.p_shard)

    def _fwd_bwd_pack(self, images, packed):
        for p in self.trainable:
            p.grad = None
        x = images if self.lowp is None else images.to(self.lowp)               # cast_forward_inputs=True
        loss, ld = self._stage_a(x, packed)
        ops.ACTIVE_PACK_PLAN = None
        self.pack.pack(0)
        return loss, ld

    def _eager_step(self, images, packed):
        loss, ld = self._fwd_bwd_pack(images, packed)
        self._reduce_scatter()
        self._update_shard()
        self._all_gather()
        return loss, ld

    # ------------------------------------------------------------------------------------------- capture / replay
    def _capture_graphs(self, g, images, packed, want_opt, capturable):
        with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
            self.loss, ld = self._fwd_bwd_pack(images, packed)
            self.scalars = ld._scalars
        self.graph = g
        gc_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gc_, pool=g.pool(), stream=self.stream, capture_error_mode="thread_local"):
            self._update_shard()
        self.graph_c = gc_
        self.pack.finish_capture()
        finish = getattr(self.optimizer, "finish_capture", None)
        if finish is not None:
            finish()
        return self

    def step(self):
        images, packed = self.static
        if self.graph is None:
            loss, ld = self._eager_step(images, packed)
            self.scalars = ld._scalars
            return loss
        sync = getattr(self.optimizer, "sync_hyper", None)
        if sync is not None:
            sync()
        self.graph.replay()
        self._reduce_scatter()
        self.graph_c.replay()
        self._all_gather()
        return self.loss

    # ------------------------------------------------------------------------------------------- checkpoints
    def full_state_dict(self):
        return self.shard.full_state_dict()
