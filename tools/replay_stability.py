import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import functions as F_
from src.model.model_builder import Model
F_.HEAD_TWO_STREAMS = False      # the per-layer hooks below read each output on the current stream right after its module
from src.model.model_blocks import Conv
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])
torch.manual_seed(0)
model = Model(**NANO, num_classes=80).cuda().train()
model.prepack = os.environ.get("PREPACK", "0") == "1"
img = torch.randn(2, 3, 160, 160, generator=torch.Generator().manual_seed(5)).cuda()
names, mods = [], []
for n, m in model.named_modules():
    if isinstance(m, Conv): names.append(n); mods.append(m)
stats = torch.zeros(len(mods), device="cuda", dtype=torch.float64)
def mk(i):
    def hook(mod, inp, out):
        stats[i] = out.detach().double().abs().sum()
    return hook
for i, m in enumerate(mods): m.register_forward_hook(mk(i))
from src.model.losses import PackedTargets, YoloDFLQFLoss
g = torch.Generator().manual_seed(6)
gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                  torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
packed = PackedTargets(gts, img.device)
crit = YoloDFLQFLoss(num_classes=80)
extra = torch.zeros(4, device="cuda", dtype=torch.float64)
def fwd():
    preds, a, s = model(img)
    extra[0] = preds.detach().double().abs().sum()
    if os.environ.get("LOSS"):
        loss, ld = crit(preds, packed, a, s)
        extra[1] = loss.detach().double()
        extra[2] = preds.detach().double().abs().sum()
    return preds
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): fwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
eager = stats.clone()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    pr = fwd()
res = []
for i in range(3):
    gr.replay(); torch.cuda.synchronize(); res.append(stats.clone()); print('replay', i, 'preds/loss/preds-after', extra.tolist()[:3], flush=True)
for i, n in enumerate(names):
    vals = [float(r[i]) for r in res]
    flag = "" if all(abs(v - vals[0]) <= 1e-6 * abs(vals[0]) for v in vals) else "   <-- differs"
    if flag or i < 3:
        print(f"{i:3d} {n:38s} eager {float(eager[i]):14.4f} replays {vals}{flag}", flush=True)
