"""Weights & Biases run set-up (reference: src/training/wandb_setup.py).  wandb is optional here: the
package is imported lazily and its absence disables logging instead of breaking the import chain."""
from datetime import datetime


def setup_wandb(config, wandb_config, args):
    try:
        import wandb
    except ImportError:
        print("[WARNING] wandb is not installed; logging disabled")
        return None
    for other in {"ddp", "fsdp", "fsdp2"} - {args.mode}:
        config.pop(other, None)
    run = wandb.init(entity=wandb_config["entity"], project=wandb_config["project_name"], config=config,
                     name=f"{args.device}_{args.mode}_{config[args.mode]['precision']}_{wandb_config['run_name']}_"
                          f"{datetime.now().strftime('%d-%m-%Y--%H:%M:%S')}")
    print("[INFO] WanDB Initialzed")
    return run
