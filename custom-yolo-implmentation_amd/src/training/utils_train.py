"""Optimizer, checkpoint and data-parallel wrappers with the reference's names and signatures
(src/training/utils_train.py).  The wrappers are torch.distributed's own DDP / FSDP1 / FSDP2 over
RCCL ("nccl" on ROCm) or gloo; the model inside runs on the HIP kernels either way because every block is a
torch.autograd.Function over plain nn.Parameters (FSDP2 shards by the same C3K2 / SPPF / PSA classes)."""
import functools
import os
from typing import Dict, Tuple, Union

import torch
import torch.optim as optim
from torch import nn
from torch.distributed.device_mesh import init_device_mesh
from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
from torch.distributed.fsdp import MixedPrecisionPolicy, ShardingStrategy, fully_shard
from torch.distributed.fsdp.fully_sharded_data_parallel import MixedPrecision
from torch.distributed.fsdp.wrap import size_based_auto_wrap_policy
from torch.nn.parallel import DistributedDataParallel as DDP

from src.model.model_blocks import C3K2, PSA, SPPF
from src.utils.common import get_num_threads

_LOWP = ("bfloat16", "float16")


def get_optimizer(model: nn.Module, lr: float, weight_decay: float, patience: int, factor: float
                  ) -> Tuple[optim.Optimizer, optim.lr_scheduler.ReduceLROnPlateau]:
    """AdamW + ReduceLROnPlateau (reference :20-36).  GPU parameters get the one-launch `HipAdamW` (same update rule,
    state names and scheduler / GradScaler / checkpoint behaviour as torch.optim.AdamW, whose default eager form issues
    several small kernels per parameter): plain parameters (single GPU, DDP) and FSDP2's DTensor parameters, whose local
    shards it updates in place.  FSDP1 (use_orig_params=True) exposes plain-looking nn.Parameters that are views of its
    flat shards, re-pointed every step: it keeps torch.optim.AdamW, and so does anything on the CPU."""
    from torch.distributed.tensor import DTensor
    native = getattr(model, "_native_shard", None)
    if native is not None:
        # `prepare_fsdp2_model(native_shard: true)`: the model is sharded HERE (flat low-precision parameters, this rank's
        # fp32 master shard) and the optimizer steps the master shard; train() drives both through ShardedStepRunner
        from src.training.fused_adamw import HipAdamW
        from src.training.sharded_step import ShardState
        if native.get("state") is None:
            native["state"] = ShardState(model, native["precision"])
        opt = HipAdamW([native["state"].master], lr=lr, weight_decay=weight_decay)
        native["optimizer"] = opt
        return opt, optim.lr_scheduler.ReduceLROnPlateau(opt, patience=patience, factor=factor)
    params = list(model.parameters())
    fsdp1 = isinstance(model, FSDP) or any(isinstance(m, FSDP) for m in model.modules())
    ok = lambda p: (type(p) is nn.Parameter and p.is_cuda) or (isinstance(p, DTensor) and p._local_tensor.is_cuda)
    if params and not fsdp1 and all(ok(p) for p in params):
        from src.training.fused_adamw import HipAdamW
        opt = HipAdamW(params, lr=lr, weight_decay=weight_decay)
    else:
        opt = optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    return opt, optim.lr_scheduler.ReduceLROnPlateau(opt, patience=patience, factor=factor)


_WRAPPER_SEGMENT = __import__("re").compile(r"(^|\.)(?:module|_fsdp_wrapped_module|_orig_mod|_checkpoint_wrapped_module)\.")


def canonical_state_dict(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Keys as the bare `Model` names them: wrapper segments that DDP ("module."), FSDP1, torch.compile or
    activation checkpointing insert are removed (whole dotted segments only), so a checkpoint written under one wrapper
    loads under any other (the reference's DDP checkpoints carry "module." and fail in its own `Model.load_weights`)."""
    out = {}
    for k, v in state.items():
        while True:
            k2 = _WRAPPER_SEGMENT.sub(r"\1", k)
            if k2 == k:
                break
            k = k2
        out[k] = v
    return out


def _is_sharded(model: nn.Module) -> bool:
    from torch.distributed.tensor import DTensor
    if getattr(model, "_native_shard", None) is not None:
        return True
    return isinstance(model, FSDP) or any(isinstance(m, FSDP) for m in model.modules()) or \
        any(isinstance(p, DTensor) for p in model.parameters())


def checkpoint_states(model: nn.Module, optimizer: optim.Optimizer):
    """(model_state, optimizer_state) to write.  Plain / DDP models: exactly what the reference writes
    (`model.state_dict()`, `optimizer.state_dict()`; :47-53).  Sharded models (FSDP1, FSDP2): FULL tensors under
    canonical names, gathered by torch.distributed.checkpoint -- a COLLECTIVE, call it on every rank (the reference
    pickles each rank's DTensor shards, which nothing can load back: notebooks/04 load error)."""
    native = getattr(model, "_native_shard", None)
    if native is not None and native.get("state") is not None:      # collectives: call on every rank
        st = native["state"]
        return st.full_state_dict(), st.full_optimizer_state_dict(optimizer)
    if not _is_sharded(model):
        return model.state_dict(), optimizer.state_dict()
    from torch.distributed.checkpoint.state_dict import StateDictOptions, get_state_dict
    return get_state_dict(model, optimizer, options=StateDictOptions(full_state_dict=True, cpu_offload=True))


def save_checkpoint(model: nn.Module, optimizer: optim.Optimizer, epoch: int, val_loss: float,
                    checkpoint_dir: str = "experiments/checkpoints", states=None) -> None:
    """{epoch, model_state, optimizer_state, val_loss} -> model_epoch_{E}.pth (reference :38-56).  `states`: the
    result of checkpoint_states() when the model is sharded (gathered on all ranks before rank 0 calls this)."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = f"{checkpoint_dir}/model_epoch_{epoch}.pth"
    model_state, opt_state = states if states is not None else checkpoint_states(model, optimizer)
    torch.save({"epoch": epoch, "model_state": model_state, "optimizer_state": opt_state, "val_loss": val_loss}, path)
    print(f"[INFO] Saved checkpoint at {path}")


def load_checkpoint(model: nn.Module, optimizer: optim.Optimizer, path: str, map_location="cpu") -> int:
    """Resume from a checkpoint written by this package or by the reference, under any wrapper: returns the epoch.
    Model keys are matched by canonical name; sharded models take full tensors through torch.distributed.checkpoint
    (collective).  The optimizer state is loaded when given and present."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    state = canonical_state_dict(ck["model_state"] if isinstance(ck, dict) and "model_state" in ck else ck)
    native = getattr(model, "_native_shard", None)
    if native is not None:
        st = native.get("state")
        if st is None:                              # before get_optimizer: plain load, sharded afterwards from these values
            model.load_state_dict(state)
        else:                                       # full tensors -> this rank's fp32 shard + every rank's compute copy
            with torch.no_grad():
                full = torch.zeros(st.total, dtype=torch.float32, device=st.master.device)
                named = dict(model.named_parameters())
                by_id = {id(q): (o, n) for q, (o, n) in zip(st.trainable, st.slices)}
                for k, v in state.items():
                    q = named.get(k)
                    if q is not None and id(q) in by_id:
                        o, n = by_id[id(q)]
                        full[o:o + n].copy_(v.reshape(-1).float())
                    elif q is not None:
                        q.copy_(v.to(q.dtype))
                for k, b in model.named_buffers():
                    if k in state:
                        b.copy_(state[k].to(b.dtype))
                lo = st.rank * st.shard_elems
                st.master.data.copy_(full[lo:lo + st.shard_elems])
                st.flat_p.copy_(full.to(st.flat_p.dtype))
            if optimizer is not None and isinstance(ck, dict) and ck.get("optimizer_state", {}).get("state"):
                os_ = ck["optimizer_state"]["state"]
                m1 = torch.zeros(st.total, dtype=torch.float32, device=st.master.device)
                m2 = torch.zeros_like(m1)
                for i, (q, (o, n)) in enumerate(zip(st.trainable, st.slices)):
                    e = os_.get(i, os_.get(str(i)))
                    if e is not None:
                        m1[o:o + n].copy_(e["exp_avg"].reshape(-1).float())
                        m2[o:o + n].copy_(e["exp_avg_sq"].reshape(-1).float())
                step = float(next(iter(os_.values()))["step"])
                lo = st.rank * st.shard_elems
                optimizer.load_state_dict(dict(
                    state={0: dict(step=torch.tensor(step), exp_avg=m1[lo:lo + st.shard_elems].clone(),
                                   exp_avg_sq=m2[lo:lo + st.shard_elems].clone())},
                    param_groups=[dict(ck["optimizer_state"]["param_groups"][0], params=[0])]))
        return int(ck["epoch"]) if isinstance(ck, dict) and "epoch" in ck else 0
    if _is_sharded(model):
        from torch.distributed.checkpoint.state_dict import StateDictOptions, set_model_state_dict, set_optimizer_state_dict
        opts = StateDictOptions(full_state_dict=True, cpu_offload=True)
        set_model_state_dict(model, state, options=opts)
        if optimizer is not None and isinstance(ck, dict) and "optimizer_state" in ck:
            set_optimizer_state_dict(model, optimizer, ck["optimizer_state"], options=opts)
    else:
        target = model.module if isinstance(model, DDP) else model
        target.load_state_dict(state)
        if optimizer is not None and isinstance(ck, dict) and "optimizer_state" in ck:
            optimizer.load_state_dict(ck["optimizer_state"])
    return int(ck["epoch"]) if isinstance(ck, dict) and "epoch" in ck else 0


def _pin_device(device: str, device_id: int, world_size: int):
    if device == "cuda":
        torch.cuda.set_device(device_id)
    else:
        torch.set_num_threads(get_num_threads(world_size))


def prepare_ddp_model(model: nn.Module, device_id: int, config: Dict[str, Union[str, int]], world_size: int,
                      device: str) -> nn.Module:
    """DistributedDataParallel: bucketed gradient all-reduce overlapped with backward (reference :167-192).  `train()`
    steps the wrapped model through `TrainStepRunner` on its `.module` by default (src/training/train_model.py
    CapturedTraining): the wrapper keeps its one-time work (parameters and buffers broadcast from rank 0), its
    state-dict naming and its eager semantics for everything outside the captured step."""
    _pin_device(device, device_id, world_size)
    model = model.to(device_id if device == "cuda" else device)
    unused = bool(config.get("find_unused_parameters", False)) if config else False
    if device != "cuda":
        return DDP(model, device_ids=None, find_unused_parameters=unused)
    # Built on the step stream: DDP keeps each parameter's AccumulateGrad node alive, and autograd runs such a node on
    # the stream it was created on -- on the default stream that would pull the default stream into the captured step
    # (hipStreamEndCapture then dies: tools/capture_probe.py).  TrainStepRunner.capture checks the streams of the nodes on its
    # warm-up step and refuses to capture otherwise; `captured_ok` only documents that this wrapper was built the right way.
    from src.hipops import functions as F_
    st = F_.step_stream(torch.device("cuda", device_id))
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ddp = DDP(model, device_ids=[device_id], find_unused_parameters=unused)
    torch.cuda.current_stream().wait_stream(st)
    ddp.captured_ok = True
    return ddp


def _fsdp1_wrap_policy(min_num_params: int):
    """size_based_auto_wrap_policy(min_num_params) restricted to modules whose `forward` this package really calls (the
    block classes of src.model: Conv, Residual, C3K, C3K2, SPPF, Attention, PSABlock, PSA, Backbone, Neck, Head).  A bare
    nn.Conv2d / nn.BatchNorm2d is a parameter container here (the enclosing block reads its parameters inside ITS forward)
    and nn.Sequential / nn.ModuleList members are iterated by the blocks (each producer writes into its slice of the
    consumer's concat buffer), so their own `forward` never runs: as FSDP units they would never be unsharded."""
    size_policy = functools.partial(size_based_auto_wrap_policy, min_num_params=int(min_num_params))

    def policy(module, recurse, nonwrapped_numel):
        if recurse:
            return True
        if not type(module).__module__.startswith("src.model."):
            return False
        return size_policy(module=module, recurse=False, nonwrapped_numel=nonwrapped_numel)
    return policy


def prepare_fsdp_model(model: nn.Module, device_id: int, config: Dict[str, Union[str, int]], world_size: int,
                       device: str) -> nn.Module:
    """FSDP1 (reference :58-114).  Two deviations, on purpose:
    * the reference compares the configured strategy with "FULLY_SHARD" (not a ShardingStrategy name), so config.yaml's
      "FULL_SHARD" silently ran NO_SHARD; here any valid ShardingStrategy name is honoured;
    * under mixed precision torch wraps every BatchNorm2d as a separate fp32 FSDP unit by default
      (`MixedPrecision._module_classes_to_ignore`).  Such a unit is unsharded only around the BatchNorm module's own
      forward, which the fused Conv block never calls; the BatchNorm parameters therefore stay in the enclosing unit and
      follow `param_dtype` / `buffer_dtype` like every other parameter -- the contract of the reference's FSDP2 path
      (:146-153); the kernels take them in that dtype and compute the normalisation in fp32 either way."""
    _pin_device(device, device_id, world_size)
    mp = None
    if config["precision"] in _LOWP:
        print("[INFO] Setting up precision - {}".format(config["precision"]))
        dt = getattr(torch, config["precision"])
        mp = MixedPrecision(param_dtype=dt, reduce_dtype=dt, buffer_dtype=dt, cast_forward_inputs=True,
                            _module_classes_to_ignore=())
    strategy = ShardingStrategy.NO_SHARD
    name = str(config.get("sharding_strategy", "NO_SHARD"))
    if world_size != 1 and name in ShardingStrategy.__members__:
        strategy = ShardingStrategy[name]
    mesh = init_device_mesh("cuda" if device == "cuda" else "cpu", (world_size,))
    policy = _fsdp1_wrap_policy(config["auto_wrap_policy_min_params"])
    return FSDP(model, auto_wrap_policy=policy, sharding_strategy=strategy, mixed_precision=mp, use_orig_params=True,
                device_id=device_id if device == "cuda" else torch.device("cpu"), device_mesh=mesh).to(device)


def prepare_fsdp2_model(model: nn.Module, device_id: int, config: Dict[str, Union[str, int]], world_size: int,
                        device: str) -> nn.Module:
    """FSDP2: fully_shard every C3K2 / SPPF / PSA, then the root (reference :116-165).

    Optional config key `native_shard: true` (GPU only; an extension, default off): no torch wrapper -- the model is marked
    for the native sharded step (src/training/sharded_step.py: the same numeric contract and the same sharding of master
    weights and optimizer state, but the parameters are gathered once per step and the step is captured), which
    `get_optimizer` and `train()` then set up.  Bit-identical to the torch-FSDP2 step in deterministic mode, 3.9x its rate
    on preset l (tests/test_gpu_fsdp.py, bench.py extra)."""
    _pin_device(device, device_id, world_size)
    model = model.to(device_id if device == "cuda" else device)
    if config.get("native_shard") and device == "cuda":
        if config.get("precision") == "float16":
            raise RuntimeError("fsdp2.native_shard: float16 is not supported (use bfloat16 or float32)")
        model._native_shard = dict(precision=config.get("precision") or "float32", state=None)
        return model
    policy = MixedPrecisionPolicy(param_dtype=None, reduce_dtype=None, cast_forward_inputs=True)
    if config.get("precision") in _LOWP:
        dt = getattr(torch, config["precision"])
        policy = MixedPrecisionPolicy(param_dtype=dt, reduce_dtype=dt, cast_forward_inputs=True)
        for buf in model.buffers():          # BN statistics follow the parameter dtype, as in the reference
            buf.data = buf.data.to(dtype=dt)
    mesh = init_device_mesh("cuda" if device == "cuda" else "cpu", (world_size,))
    for module in reversed(list(model.modules())):
        if isinstance(module, (C3K2, SPPF, PSA)):
            fully_shard(module, mp_policy=policy, reshard_after_forward=True, mesh=mesh)
    fully_shard(model, mp_policy=policy, reshard_after_forward=True, mesh=mesh)
    return model
