#!/usr/bin/env python3
"""Drop-in for the reference's scripts/distributed_training.py (same CLI, same call sequence, same config.yaml
keys), running the MI355X-native model:

    torchrun --nnodes=1 --nproc-per-node N scripts/distributed_training.py --device cuda --mode {ddp,fsdp,fsdp2} \
             --precision {float32,bfloat16,float16} [--batch_size B] [--load_from_checkpoint DIR]

`--device cpu` is refused: the product path has no CPU fallback (the CPU oracle under oracle/ is the checker).
wandb / torchinfo are optional.  Unlike the reference, a failure exits non-zero instead of being swallowed."""
import argparse
import json
import os
import sys
from datetime import datetime

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

from src.data.data_loader import get_data_loaders  # noqa: E402
from src.model.losses import YoloDFLQFLoss  # noqa: E402
from src.model.model_builder import Model  # noqa: E402
from src.training.distributed_setup import cleanup_distribute_mode, init_distributed_mode  # noqa: E402
from src.training.train_model import train  # noqa: E402
from src.training.utils_train import (get_optimizer, load_checkpoint, prepare_ddp_model, prepare_fsdp2_model,  # noqa: E402
                                      prepare_fsdp_model)
from src.training.wandb_setup import setup_wandb  # noqa: E402
from src.utils.common import find_latest_checkpoint, get_checkpoint_config  # noqa: E402
from src.utils.config_loader import load_config  # noqa: E402

WRAP = {"ddp": prepare_ddp_model, "fsdp": prepare_fsdp_model, "fsdp2": prepare_fsdp2_model}


def main(args):
    if args.device != "cuda":
        raise SystemExit("--device cpu: this build runs the hot path on MI355X only (no CPU fallback)")
    cfg = load_config(args.config)
    data_cfg, tr_cfg, model_cfg, ck_cfg = cfg["data"], cfg["training"], cfg["model"], cfg["checkpoint"]
    root = ck_cfg.get("checkpoint_dir", "experiments/checkpoints")
    initial_epoch = 0
    if args.load_from_checkpoint:
        ckpt_dir = os.path.join(root, args.load_from_checkpoint)
        saved = get_checkpoint_config(ckpt_dir)
        model_cfg["config"], model_cfg["num_classes"] = saved["config"], saved["num_classes"]
        args.precision, args.mode = saved["precision"], saved["mode"]
        print(f"[INFO] Loaded model config from checkpoint directory: precision = {args.precision}, mode = {args.mode}")
    else:
        ckpt_dir = os.path.join(root, datetime.now().strftime("%d-%m-%Y--%H-%M-%S"))
        os.makedirs(ckpt_dir, exist_ok=True)
        with open(os.path.join(ckpt_dir, "model_config.json"), "w") as fh:
            json.dump({"config": model_cfg["config"], "num_classes": model_cfg.get("num_classes", 172),
                       "mode": args.mode, "precision": args.precision}, fh)
        print("[INFO] Model config saved to checkpoint directory")

    rank, world_size, gpu = init_distributed_mode(device=args.device)
    use_wandb = cfg.get("wandb", {}).get("enable", False)
    run = None
    try:
        tr_cfg[args.mode]["precision"] = args.precision
        if args.batch_size is not None:
            tr_cfg["batch_size"] = args.batch_size
        if args.prefetch_factor is not None:
            data_cfg["prefetch_factor"] = args.prefetch_factor
        if rank == 0 and use_wandb:
            run = setup_wandb(config={"device": args.device, "world_size": world_size, "mode": args.mode,
                                      "checkpoint_path": ckpt_dir, **tr_cfg}, wandb_config=cfg["wandb"], args=args)
        use_wandb = use_wandb and run is not None

        model = Model(**model_cfg["config"], num_classes=model_cfg["num_classes"])
        # the training step as replayed hipGraphs: default in ddp mode on one GPU, opt-in with more ranks (optional key
        # `training.captured_step`: true = everywhere, false = the reference's eager loop); optional key
        # `training.ddp.grad_compress: bf16` = bf16 gradient buckets in the captured step (default: fp32, as the reference)
        captured = tr_cfg.get("captured_step", None)
        captured = None if captured is None else bool(captured)
        model = WRAP[args.mode](model=model, device_id=gpu, config=tr_cfg[args.mode], world_size=world_size, device=args.device)
        print(f"[INFO] {args.mode.upper()} model initialzed")
        model = model.to(args.device)

        train_loader, val_loader = get_data_loaders(
            train_parquet=os.path.join(data_cfg["processed_dir"], data_cfg["train_parquet"]),
            val_parquet=os.path.join(data_cfg["processed_dir"], data_cfg["val_parquet"]),
            train_images=data_cfg["train_images"], val_images=data_cfg["val_images"], batch_size=tr_cfg["batch_size"],
            is_test=tr_cfg["is_test"], prefetch_factor=data_cfg.get("prefetch_factor", 2), percent=args.dataset_percent,
            device=args.device, num_classes=model_cfg["num_classes"])
        optimizer, scheduler = get_optimizer(model=model, lr=tr_cfg["learning_rate"], weight_decay=tr_cfg["weight_decay"],
                                             patience=tr_cfg["learning_rate_patience"], factor=tr_cfg["learning_rate_factor"])
        if args.load_from_checkpoint:
            path = find_latest_checkpoint(ckpt_dir)
            initial_epoch = load_checkpoint(model, optimizer, path, map_location=args.device)
            print(f"[INFO] Loaded model and optimizer from checkpoint at epoch {initial_epoch} from {path}")
        criterion = YoloDFLQFLoss(num_classes=model_cfg["num_classes"], lambda_box=tr_cfg["weights"].get("bbox_loss", 1.5),
                                  lambda_cls=tr_cfg["weights"].get("cls_loss", 1.0))
        train(model=model, train_loader=train_loader, val_loader=val_loader, optimizer=optimizer, scheduler=scheduler,
              criterion=criterion, initial_epoch=initial_epoch, num_epochs=initial_epoch + tr_cfg["epochs"], device=gpu,
              num_classes=model_cfg["num_classes"], rank=rank, use_wandb=use_wandb, wandb_instance=run,
              log_interval=tr_cfg.get("log_interval", 10), checkpoint_dir=ckpt_dir,
              iou_threshold=tr_cfg.get("iou_threshold", 0.5), conf_threshold=tr_cfg.get("conf_threshold", 0.25),
              distributed_mode=args.mode, precision=args.precision, captured_step=captured,
              grad_compress=(tr_cfg.get("ddp") or {}).get("grad_compress"))
    finally:
        if run is not None:
            import wandb
            wandb.finish()
            print("[INFO] WanDB destroyed")
        cleanup_distribute_mode()


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Disrtibuted training with FSDP or DDP.")
    ap.add_argument("--device", type=str, default="cuda", choices=["cpu", "cuda"], metavar="D")
    ap.add_argument("--mode", type=str, required=True, choices=["fsdp", "ddp", "fsdp2"], metavar="M")
    ap.add_argument("--precision", type=str, default="float32", choices=["bfloat16", "float16", "float32"], metavar="P")
    ap.add_argument("--batch_size", type=int, default=None, metavar="B")
    ap.add_argument("--prefetch_factor", type=int, default=None, metavar="F")
    ap.add_argument("--dataset_percent", type=float, default=1.0, metavar="DP")
    ap.add_argument("--load_from_checkpoint", type=str, default=None, metavar="LC")
    ap.add_argument("--config", type=str, default="config.yaml", help="path of the YAML configuration (extension)")
    main(ap.parse_args())
