"""AdamW over every parameter in ONE kernel launch: `HipAdamW`, a drop-in for the
`torch.optim.AdamW` that the reference's get_optimizer builds (src/training/utils_train.py:34; step at
src/training/train_model.py:247-253).

* same update rule, defaults, `param_groups` keys and per-parameter state names (`step`, `exp_avg`, `exp_avg_sq`),
  so `ReduceLROnPlateau`, checkpoints (`state_dict` / `load_state_dict`) and `GradScaler.step` work unchanged;
* hyper-parameters and the step counter live in device memory: a captured step follows a learning-rate schedule
  without recapture (`sync_hyper()` copies changed values; `step()` calls it when not capturing);
* the kernel reads a device job table (parameter / gradient / moment pointers).  Gradient tensors are re-created by
  autograd every eager step, so the table is rebuilt when a pointer changed; inside a graph capture the pointers are
  final but an upload cannot be captured safely, so `step()` only records the launch and `finish_capture()` -- called
  by TrainStepRunner after the capture -- uploads the table once.
* FSDP2 (`fully_shard`) parameters are DTensors: the kernel updates each rank's LOCAL shard (a persistent view of the
  wrapper's sharded storage) from the local shard of the reduce-scattered gradient; the moments are DTensors with the
  parameter's placement, so `torch.distributed.checkpoint` gathers / scatters the optimizer state like torch.optim.AdamW's.
  There GradScaler unscales the gradients itself (`_step_supports_amp_scaling` off: ShardedGradScaler must all-reduce
  found_inf across ranks before anyone steps).
There is no CPU path: parameters must live on the GPU (like every op of this package).
"""
import torch

from src.hipops import lib
from src.hipops.ops import _p, _stream, dt

try:
    from torch.distributed.tensor import DTensor
except ImportError:  # pragma: no cover
    DTensor = ()


def _loc(t):
    """The tensor the kernel touches: a DTensor's local shard, else the tensor itself."""
    return t._local_tensor if isinstance(t, DTensor) else t


def _grad(p):
    """The gradient the kernel reads: `p.lowp_grad` when set (a bf16 / f16 gradient of an fp32 master parameter -- torch
    refuses such a tensor as `.grad`; ShardedStepRunner's reduce-scattered shard), else `p.grad`."""
    g = getattr(p, "lowp_grad", None)
    return g if g is not None else p.grad


class DeviceGradScaler:
    """torch.amp.GradScaler's state and rules (defaults: init_scale 65536, growth 2, backoff 0.5, interval 2000 -- what the
    reference constructs, src/training/train_model.py:195-208) held in device memory, for a training step that never
    returns to the host: `scale` multiplies the loss gradient inside the loss kernel (functions.LOSS_SCALE), `HipAdamW.step`
    with `optimizer.device_amp = this` checks every gradient for inf / nan, skips or applies the unscaled update and
    updates the scale (csrc/optim.hip: yolo_adamw_amp_step).  One optimizer with ONE parameter group per scaler."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0], dtype=torch.float32, device=device)   # scale, found_inf, last found_inf
        self.tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    @property
    def scale(self):
        """fp32 device scalar (a view: the kernels update it in place)."""
        return self.state[0:1]

    def get_scale(self):
        return float(self.state[0])                 # host sync: for logging / tests only

    def last_step_skipped(self):
        return bool(self.state[2] != 0)             # host sync: for logging / tests only

    def state_dict(self):
        return dict(scale=self.get_scale(), growth_tracker=int(self.tracker), growth_factor=self.growth_factor,
                    backoff_factor=self.backoff_factor, growth_interval=self.growth_interval)

    def load_state_dict(self, sd):
        self.state[0] = float(sd["scale"])
        self.tracker.fill_(int(sd.get("growth_tracker", sd.get("_growth_tracker", 0))))
        self.growth_factor, self.backoff_factor = sd["growth_factor"], sd["backoff_factor"]
        self.growth_interval = sd["growth_interval"]


class HipAdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True      # GradScaler hands over grad_scale / found_inf instead of unscaling itself

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=True):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=True, differentiable=False, fused=True)
        super().__init__(params, defaults)
        if any(isinstance(p, DTensor) for g in self.param_groups for p in g["params"]):
            self._step_supports_amp_scaling = False     # sharded: the scaler unscales and agrees on found_inf across ranks
        self._plans = {}                    # group index -> dict(ptrs, jobs_dev, njobs, nchunks, hyper, hyper_host, step)
        self._pending = []                  # (jobs_dev, pinned host table) awaiting upload after a capture
        # grad_scale / found_inf are NOT pre-defined: GradScaler.step multiplies an existing grad_scale attribute in,
        # sets both around step() and deletes them afterwards

    # ------------------------------------------------------------------------------------------ state
    def _init_state(self, group, gi):
        plan = self._plans.get(gi)
        if plan is None:
            dev = next(p.device for p in group["params"])
            plan = self._plans[gi] = dict(ptrs=None, jobs_dev=None, host=None, njobs=0, nchunks=0, hyper_host=None,
                                          hyper=torch.zeros(5, dtype=torch.float64, device=dev),
                                          step=torch.zeros((), dtype=torch.float32, device=dev))
        for p in group["params"]:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = plan["step"]                           # one shared device counter per group
                if isinstance(p, DTensor):                           # same mesh / placement as the parameter
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
                else:
                    st["exp_avg"] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
                    st["exp_avg_sq"] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
            elif st["step"] is not plan["step"]:                    # after load_state_dict: adopt the loaded count
                plan["step"].copy_(torch.as_tensor(st["step"], dtype=torch.float32).reshape(()))
                st["step"] = plan["step"]
                st["exp_avg"] = st["exp_avg"].to(torch.float32)
                st["exp_avg_sq"] = st["exp_avg_sq"].to(torch.float32)
        return plan

    def sync_hyper(self):
        """Copy changed hyper-parameters (e.g. a scheduler's new lr) to the device; call between graph replays."""
        for gi, group in enumerate(self.param_groups):
            plan = self._plans.get(gi)
            if plan is None:
                continue
            want = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                    float(group["weight_decay"]))
            if plan["hyper_host"] != want:
                plan["hyper"].copy_(torch.tensor(want, dtype=torch.float64))
                plan["hyper_host"] = want

    def finish_capture(self):
        """Upload the job tables recorded while a graph was being captured (call once after the capture ends)."""
        for dev_t, host_t in self._pending:
            dev_t.copy_(host_t)
        self._pending = []
        for plan in self._plans.values():               # kept for restore_capture()
            if plan["host"] is not None:
                plan["cap_host"], plan["cap_ptrs"] = plan["host"].clone(), plan["ptrs"]
        self.sync_hyper()

    def restore_capture(self):
        """After an EAGER step() between replays (its gradients live elsewhere, so it rebuilt the job table the captured
        launch reads): put the captured table back."""
        for plan in self._plans.values():
            if plan.get("cap_host") is not None and plan["ptrs"] != plan["cap_ptrs"]:
                plan["host"].copy_(plan["cap_host"])
                plan["jobs_dev"].copy_(plan["host"])
                plan["ptrs"] = plan["cap_ptrs"]

    # ------------------------------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        """One launch for all parameters of a group.  Deviation from torch.optim.AdamW: the step counter (bias correction)
        is per GROUP, not per parameter -- a parameter that receives no gradient on some steps is corrected as if it had
        been stepped with the others (the reference's model gives every parameter a gradient on every step)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if _grad(p) is not None]
            if not params:
                continue
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("HipAdamW implements plain AdamW (amsgrad=False, maximize=False)")
            plan = self._init_state(group, gi)
            capturing = torch.cuda.is_current_stream_capturing()
            ptrs = tuple((_loc(p).data_ptr(), _loc(_grad(p)).data_ptr(), _loc(p).numel()) for p in params)
            if plan["ptrs"] != ptrs:
                old = plan["ptrs"]
                if (not capturing and old is not None and len(old) == len(ptrs) and plan["host"] is not None
                        and all(a[0] == b[0] and a[2] == b[2] for a, b in zip(old, ptrs))
                        and plan.get("gdtypes") == tuple(_grad(p).dtype for p in params)):
                    # same parameters, fresh gradient tensors (an eager loop's zero_grad(set_to_none=True)): one call
                    # rewrites the gradient pointers instead of a job_fill call per parameter
                    # The upload is asynchronous from one of TWO pinned staging tables; a table is rewritten only after
                    # the upload that last read it has completed (its event): no host sync unless the device is two
                    # steps behind.
                    import ctypes
                    stage = plan.setdefault("stage", [plan["host"].clone().pin_memory(), plan["host"].clone().pin_memory()])
                    evs = plan.setdefault("stage_ev", [None, None])
                    k = plan["stage_k"] = 1 - plan.get("stage_k", 1)
                    if evs[k] is not None:
                        evs[k].synchronize()
                    stage[k].copy_(plan["host"])
                    arr = (ctypes.c_void_p * len(ptrs))(*[t[1] for t in ptrs])
                    lib.call("yolo_adamw_jobs_set_grads", stage[k].data_ptr(), len(ptrs), arr)
                    plan["jobs_dev"].copy_(stage[k], non_blocking=True)
                    evs[k] = torch.cuda.Event()
                    evs[k].record()
                else:
                    self._build(plan, params, capturing)
                    plan["gdtypes"] = tuple(_grad(p).dtype for p in params)
                plan["ptrs"] = ptrs
            if not capturing:
                self.sync_hyper()
            elif plan["hyper_host"] is None:
                raise RuntimeError("HipAdamW: run one eager step (or sync_hyper()) before capturing a graph")
            amp = getattr(self, "device_amp", None)
            if amp is not None:     # fp16 loss scaling kept on the device (DeviceGradScaler): found_inf, step, scale update
                lib.call("yolo_adamw_amp_step", _p(plan["jobs_dev"]), plan["njobs"], plan["nchunks"], _p(plan["hyper"]),
                         _p(plan["step"]), _p(amp.state), _p(amp.tracker), float(amp.growth_factor), float(amp.backoff_factor),
                         int(amp.growth_interval), _stream(_loc(params[0])))
                continue
            lib.call("yolo_adamw_step", _p(plan["jobs_dev"]), plan["njobs"], plan["nchunks"], _p(plan["hyper"]),
                     _p(plan["step"]), _p(getattr(self, "grad_scale", None)), _p(getattr(self, "found_inf", None)),
                     _stream(_loc(params[0])))      # GradScaler sets the two attributes around step() and deletes them after
        return loss

    def _build(self, plan, params, capturing):
        jb = lib.query("yolo_adamw_job_bytes")
        n = len(params)
        if plan["host"] is None or plan["host"].numel() != n * jb:
            if capturing:
                raise RuntimeError("HipAdamW: run one eager step before capturing a graph (host/device tables are "
                                   "allocated there)")
            plan["host"] = torch.zeros(n * jb, dtype=torch.uint8).pin_memory()
            plan["jobs_dev"] = torch.empty(n * jb, dtype=torch.uint8, device=_loc(params[0]).device)
        host = plan["host"]
        for i, p in enumerate(params):
            st = self.state[p]
            w, g, m1, m2 = _loc(p), _loc(_grad(p)), _loc(st["exp_avg"]), _loc(st["exp_avg_sq"])
            if not (w.is_contiguous() and g.is_contiguous() and m1.is_contiguous() and m2.is_contiguous()):
                raise RuntimeError("HipAdamW needs contiguous parameters, gradients and moments")
            if not (g.numel() == m1.numel() == m2.numel() == w.numel()):
                raise RuntimeError("HipAdamW: gradient / moment shards do not match the parameter's local shard")
            lib.call("yolo_adamw_job_fill", host.data_ptr(), i, _p(w), dt(w), _p(g), dt(g), _p(m1), _p(m2), w.numel())
        plan["nchunks"] = lib.query("yolo_adamw_jobs_finalize", host.data_ptr(), n)
        plan["njobs"] = n
        if capturing:
            # the pointers recorded now are the capture's final ones; the upload itself must not be captured
            # (and is not needed until the first replay): finish_capture() does it
            self._pending.append((plan["jobs_dev"], host))
        else:
            plan["jobs_dev"].copy_(host)
