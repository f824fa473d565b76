"""Which ATen ops does one bf16 forward + backward under prepare_fsdp2_model launch, and from where?  (host-side profile)"""
import os, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch, torch.distributed as dist
from torch.profiler import ProfilerActivity, profile
from oracle import blocks as ob
from oracle.params import det_fill_
from src.model.model_builder import Model
from src.training.utils_train import prepare_fsdp2_model

os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
preset = sys.argv[1] if len(sys.argv) > 1 else "n"
m = Model(**ob.PRESETS[preset], num_classes=80); det_fill_(m.state_dict(), 3)
m = prepare_fsdp2_model(model=m, device_id=0, config={"precision": "bfloat16"}, world_size=1, device="cuda").train()
img = torch.randn(2, 3, 160, 160).cuda()
for _ in range(2):
    m.zero_grad(set_to_none=True)
    m(img)[0].float().square().mean().backward()
m.zero_grad(set_to_none=True)
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    m(img)[0].float().square().mean().backward()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::"):
        frames = [f for f in (e.stack or []) if "custom-yolo" in f or "fsdp" in f.lower()]
        cnt[(e.name, frames[0].split("/")[-1] if frames else "?")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:6d} {name:34s} {where}")
print("total top-level aten calls", sum(cnt.values()))
dist.destroy_process_group()
