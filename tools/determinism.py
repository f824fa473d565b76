"""Replay determinism of the captured step (lr = 0): the loss of every replay must agree to atomics noise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.model.losses import PackedTargets, YoloDFLQFLoss
from src.model.model_builder import Model
from src.training.graph_step import TrainStepRunner
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
S = dict(csp=[False, True], depth=[1] * 6, width=[3, 32, 64, 128, 256, 512])
for name, cfg, res, n, prec in [("nano", NANO, 160, 2, "bfloat16"), ("nano", NANO, 160, 2, "float32"), ("nano", NANO, 320, 4, "bfloat16"), ("s", S, 320, 4, "bfloat16")]:
    for graph in (False, True):
        torch.manual_seed(0)
        model = Model(**cfg, num_classes=80).cuda().train()
        g = torch.Generator().manual_seed(5)
        img = torch.randn(n, 3, res, res, generator=g).cuda()
        gts = [torch.cat([torch.rand(3, 2, generator=g) * res, torch.rand(3, 2, generator=g) * 60 + 8,
                          torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(n)]
        opt = torch.optim.AdamW(model.parameters(), lr=0.0, weight_decay=0.0, capturable=True, fused=True)
        r = TrainStepRunner(model, YoloDFLQFLoss(num_classes=80), opt, prec, use_graph=graph)
        r.capture(img, PackedTargets(gts, img.device), warmup=2)
        out = []
        for i in range(5):
            loss = r.step(); torch.cuda.synchronize()
            gn = sum(float(p.grad.float().norm()) for p in model.parameters() if p.grad is not None)
            out.append((round(float(loss), 5), round(gn, 3)))
        print(name, res, prec, "graph" if graph else "eager", out, flush=True)
