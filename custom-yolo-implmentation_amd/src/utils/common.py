"""Host-side helpers with the reference's names (src/utils/common.py): worker/thread counts, checkpoint lookup."""
import glob
import json
import multiprocessing
import os

import torch


def get_num_workers():
    """DataLoader workers: SLURM_CPUS_PER_TASK // #GPUs when under SLURM, else cpu_count // #GPUs; capped at 16."""
    gpus = torch.cuda.device_count() or 1
    slurm = os.getenv("SLURM_CPUS_PER_TASK")
    n = max(1, int(slurm) // gpus) if slurm is not None else max(2, multiprocessing.cpu_count() // gpus)
    return min(n, 16)


def get_num_threads(world_size: int):
    """CPU threads per process: all cores for one process, cores // world_size otherwise."""
    if world_size > 1:
        n = max(1, multiprocessing.cpu_count() // world_size)
        print(f"Number of threads: {n}")
        return n
    print(f"Number of threads (default): {torch.get_num_threads()}")
    return torch.get_num_threads()


def get_checkpoint_config(checkpoint_path):
    path = os.path.join(checkpoint_path, "model_config.json")
    if not os.path.exists(path):
        print("[WARNING] Model config file not found in checkpoint directory")
        raise FileNotFoundError("Model config file not found in checkpoint directory")
    with open(path, "r") as fh:
        return json.load(fh)


def find_latest_checkpoint(checkpoint_dir, extension="*.pth"):
    files = [f for f in glob.glob(os.path.join(checkpoint_dir, extension)) if os.path.isfile(f)]
    if not files:
        raise FileNotFoundError(f"No checkpoint files found in directory: {checkpoint_dir}")
    return max(files, key=os.path.getmtime)
