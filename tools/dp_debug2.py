import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch, torch.distributed as dist
from src.model.losses import PackedTargets, YoloDFLQFLoss
from src.model.model_builder import Model
from src.training.graph_step import TrainStepRunner
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29514")
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0, device_id=torch.device("cuda", 0))
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
for trial in range(3):
    for force in (False, True):
        torch.manual_seed(0)
        model = Model(**NANO, num_classes=80).cuda().train()
        g = torch.Generator().manual_seed(5)
        img = torch.randn(2, 3, 160, 160, generator=g).cuda()
        gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                          torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True, fused=True)
        r = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, "bfloat16", use_graph=True, force_comm=force)
        r.capture(img, PackedTargets(gts, img.device), warmup=2)
        out = []
        for i in range(6):
            loss = r.step(); torch.cuda.synchronize()
            out.append(round(float(loss), 4))
        print("trial", trial, "two-graph" if force else "one-graph", out, flush=True)
dist.destroy_process_group()
