"""Where does a model with fp16 PARAMETERS (FSDP mixed-precision contract) leave the fp16-autocast model?  First diverging Conv."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from oracle import blocks as ob
from oracle.params import det_fill_
from src.model.model_builder import Model
from src.model.model_blocks import Conv

dt = getattr(torch, sys.argv[1] if len(sys.argv) > 1 else "float16")
cfg = ob.PRESETS["n"]
img = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(9)).cuda()

def run(mode):
    m = Model(**cfg, num_classes=80); det_fill_(m.state_dict(), 2); m = m.cuda().train()
    outs = {}
    def hook(name):
        def f(mod, inp, out):
            outs[name] = out.detach().float().clone()
        return f
    for k, mod in m.named_modules():
        if type(mod) is Conv:
            mod.register_forward_hook(hook(k))
    if mode == "fsdp2":
        from src.training.utils_train import prepare_fsdp2_model
        m = prepare_fsdp2_model(model=m.cpu(), device_id=0, config={"precision": sys.argv[1] if len(sys.argv) > 1 else "float16"}, world_size=1, device="cuda").train()
        p = m(img)[0]
        return outs, p.float()
    if mode == "fp32":
        p = m(img)[0]
        return outs, p.float()
    if mode == "amp":
        with torch.autocast("cuda", dtype=dt):
            p = m(img)[0]
    else:
        m = m.to(dt)
        p = m(img.to(dt))[0]
    return outs, p.float()

import torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
a, pa = run("fp32")
b, pb = run(sys.argv[2] if len(sys.argv) > 2 else "param")
print("preds diff", float((pa - pb).abs().max() / pa.abs().max()))
for k in a:
    d = float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-9))
    flag = " <<<" if d > 5e-2 else ""
    print(f"{k:30s} {tuple(a[k].shape)} rel diff {d:.3e} finite {bool(torch.isfinite(b[k]).all())}{flag}")
dist.destroy_process_group()
