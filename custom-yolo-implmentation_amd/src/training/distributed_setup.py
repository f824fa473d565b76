"""Process-group plumbing with the reference's names (src/training/distributed_setup.py).
On PyTorch-ROCm backend "nccl" IS RCCL (xGMI inside a node); CPU runs use gloo."""
import os

import torch
import torch.distributed as dist


def init_distributed_mode(device: str):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        rank, world_size, gpu = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    else:
        print("[WARNING] Not using distributed mode")
        rank, world_size, gpu = 0, 1, 0
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    if device == "cuda":
        torch.cuda.set_device(gpu)
    dist.init_process_group(backend="nccl" if device == "cuda" else "gloo", init_method="env://",
                            world_size=world_size, rank=rank)
    if device == "cuda":
        dist.barrier(device_ids=[gpu])
    else:
        dist.barrier()
    print("[INFO] Distributed process group initialized")
    return rank, world_size, gpu


def reduce_values(values, average=True):
    """All-reduce several Python scalars in ONE collective (the reference issues one per scalar, six per epoch)."""
    if not dist.is_initialized() or dist.get_world_size() < 2:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    if average:
        t /= dist.get_world_size()
    return t.tolist()


def reduce_value(value, average=True):
    """Reference signature (:28-63): mean (or sum) of a scalar / tensor over all ranks, returned as a float."""
    if not dist.is_initialized() or dist.get_world_size() < 2:
        return value
    if isinstance(value, torch.Tensor):
        with torch.no_grad():
            v = value.cuda() if dist.get_backend() == "nccl" and not value.is_cuda else value
            dist.all_reduce(v)
            if average:
                v /= dist.get_world_size()
            return v.item()
    return reduce_values([value], average)[0]


def cleanup_distribute_mode():
    if dist.is_initialized():
        dist.destroy_process_group()
        print("[INFO] Distirbuted process group destroyed")
