"""k_dgrad2_patch (stride-2 3x3 data gradient, 128 -> 128 @160x160) against ATen on ALL images, repeated, optionally beside a
bandwidth-hungry kernel on a second stream.  Wrote the note in conv_up2.hip: with the 16-byte epilogue path the last parity
class came out wrong in a few waves of some runs; the 8-byte path must never."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
import torch
from src.hipops import ops
torch.manual_seed(0)
n, cin, cout, h, w = int(os.environ.get("DBG_N", "32")), 128, 128, 160, 160
reps, load = int(os.environ.get("DBG_REPS", "10")), os.environ.get("DBG_LOAD", "1") == "1"
dy = torch.randn(n, cout, 80, 80).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.03
wb = ops.pack_weights(wt, 3, 2, 1, torch.bfloat16)
ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.to(torch.bfloat16).float(), dy.float(), 2, 1)
lim = 0.02 * ref.abs().max()
side = torch.cuda.Stream()
a, b = torch.empty(64 << 20, device="cuda"), torch.empty(64 << 20, device="cuda")
tot = 0
for r in range(reps):
    if load:
        with torch.cuda.stream(side):
            for _ in range(4):
                b.copy_(a)
    got = ops.conv_dgrad(dy, wb, cin, h, w, 3, 2).float()
    bad = (got - ref).abs() > lim
    nb = int(bad.sum())
    tot += nb
    if nb:
        idx = bad.nonzero()
        i0 = idx[0].tolist()
        v = got[i0[0], :, i0[2], i0[3]]                                  # 128 channels of the first bad pixel
        d = (ref[i0[0]] - v[:, None, None]).abs().amax(0)                # max channel distance to every pixel of the image
        m = int(d.argmin()); my, mx = m // w, m % w
        # channel-wise: which channels are wrong at that pixel, and do they equal the reference at some other channel block?
        wrong = ((v - ref[i0[0], :, i0[2], i0[3]]).abs() > lim).nonzero().flatten().tolist()
        print(f"  first bad pixel {i0[0]},{i0[2]},{i0[3]}: {len(wrong)} wrong channels {wrong[:12]}...; nearest reference pixel ({my},{mx}) max-dist {float(d.flatten()[m]):.3f}")
        print("   got", [round(float(t), 3) for t in v[:8]], "ref", [round(float(t), 3) for t in ref[i0[0], :8, i0[2], i0[3]]])
        print(f"rep {r}: bad {nb}; images {idx[:, 0].unique().tolist()} row parity {sorted(set((idx[:, 2] % 2).tolist()))} col parity {sorted(set((idx[:, 3] % 2).tolist()))}")
torch.cuda.synchronize()
print(f"{reps} runs, concurrent load {load}: {tot} bad elements in total")
