import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.model.losses import PackedTargets, YoloDFLQFLoss
from src.model.model_builder import Model
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
torch.manual_seed(0)
model = Model(**NANO, num_classes=80).cuda().train()
model.prepack = os.environ.get("PREPACK", "1") == "1"
g = torch.Generator().manual_seed(5)
img = torch.randn(2, 3, 160, 160, generator=g).cuda()
gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                  torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
packed = PackedTargets(gts, img.device)
crit = YoloDFLQFLoss(num_classes=80)
prec = os.environ.get("PREC", "float32")
def fwd_bwd(do_bwd):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=prec == "bfloat16"):
        preds, a, s = model(img)
        loss, ld = crit(preds, packed, a, s)
    if do_bwd:
        loss.backward()
    return loss, preds
def gnorm():
    return round(sum(float(p.grad.float().norm()) for p in model.parameters() if p.grad is not None), 4)
# eager reference
for p in model.parameters(): p.grad = None
l, pr = fwd_bwd(True); torch.cuda.synchronize()
print("eager      loss", float(l), "preds", float(pr.float().abs().sum()), "gnorm", gnorm(), flush=True)
for do_bwd in (False, True):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for p in model.parameters(): p.grad = None
            fwd_bwd(do_bwd)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    for p in model.parameters(): p.grad = None
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        l, pr = fwd_bwd(do_bwd)
    for i in range(4):
        gr.replay(); torch.cuda.synchronize()
        print("graph bwd=%d replay %d loss" % (do_bwd, i), float(l), "preds", float(pr.float().abs().sum()), "gnorm", gnorm() if do_bwd else None, flush=True)
