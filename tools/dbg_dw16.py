import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from src.hipops import ops
import emulated_ops as emu
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    for c, h, w in ((16, 12, 12), (128, 10, 10), (64, 40, 40)):
        x = torch.randn(2, c, h, w).to(dt).contiguous(memory_format=torch.channels_last)
        w9 = torch.randn(c, 9) * 0.3
        ref = emu.dw_fwd(x, w9).float()
        for stats in (False, True):
            acc = ops.bn_acc_new(c, "cuda") if stats else None
            y = ops.dw_fwd(x.cuda(), w9.cuda(), acc).float().cpu()
            e = float((y - ref).abs().max() / ref.abs().max())
            print(dt, (c, h, w), "stats" if stats else "plain", f"rel err {e:.3e}", "y[0,:4,0,0]", y[0, :4, 0, 0].tolist(), "ref", ref[0, :4, 0, 0].tolist())
        dy = torch.randn(2, c, h, w).to(dt).contiguous(memory_format=torch.channels_last)
        dx = ops.dw_dgrad(dy.cuda(), w9.cuda()).float().cpu()
        rdx = emu.dw_dgrad(dy, w9).float()
        dw = ops.dw_wgrad(x.cuda(), dy.cuda()).float().cpu()
        rdw = emu.dw_wgrad(x, dy).float()
        print(dt, (c, h, w), f"dgrad rel err {float((dx - rdx).abs().max() / rdx.abs().max()):.3e}  wgrad rel err {float((dw - rdw).abs().max() / rdw.abs().max()):.3e}")
