import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops
torch.manual_seed(0)
n, cin, cout, h, w = int(os.environ.get("DBG_N", "32")), 128, 128, 160, 160
dy = (torch.randn(n, cout, 80, 80) ).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.03
wb = ops.pack_weights(wt, 3, 2, 1, torch.bfloat16)
got = ops.conv_dgrad(dy, wb, cin, h, w, 3, 2).float()
ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.to(torch.bfloat16).float(), dy.float(), 2, 1)
bad = (got - ref).abs() > 0.02 * ref.abs().max()
print("bad", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
if len(idx):
    print("images", idx[:, 0].unique().tolist())
    print("channels", idx[:, 1].unique().tolist()[:40])
    print("rows", idx[:, 2].unique().tolist()[:40])
    print("cols", idx[:, 3].unique().tolist()[:40])
    # channel-quads pattern of first bad pixel
    i0 = idx[0]
    print("first", i0.tolist(), "bad channels at that pixel:", bad[i0[0], :, i0[2], i0[3]].nonzero().flatten().tolist())
    print("got", got[i0[0], :16, i0[2], i0[3]].tolist()); print("ref", ref[i0[0], :16, i0[2], i0[3]].tolist())
