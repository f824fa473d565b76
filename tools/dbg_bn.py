import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops
y = torch.randn(2, 64, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
acc = ops.bn_acc_new(64, "cuda")
ops.bn_stats_acc(y, acc)
torch.cuda.synchronize(); print("stats ok", flush=True)
g = torch.ones(64, device="cuda"); b = torch.zeros(64, device="cuda"); rm = torch.zeros(64, device="cuda"); rv = torch.ones(64, device="cuda")
m, i, s, sh = ops.bn_finalize_acc(acc, 128, g, b, rm, rv, 0.03, 1e-3)
torch.cuda.synchronize(); print("finalize_acc ok", flush=True)
sc, sf = ops.bn_eval_coeffs(g, b, rm, rv, 1e-3)
torch.cuda.synchronize(); print("eval ok", flush=True)
out = ops.bn_act_fwd_train(y, acc, g, b, rm, rv, 0.03, 1e-3, 1)
torch.cuda.synchronize(); print("fwd_train ok", flush=True)
