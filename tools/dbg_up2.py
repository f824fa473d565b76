"""k_dgrad2_patch (stride-2 3x3 data gradient, 128 -> 128 @160x160) against ATen on ALL images, repeated, optionally beside a
bandwidth-hungry kernel on a second stream, with histograms of where in a workgroup tile the wrong elements sit.  The
reproducer of round 3's ring race: YOLO_CONV_WIDE=0/1/2 selects the epilogue's store / exchange form."""
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
import collections
hist = {k: collections.Counter() for k in ("dy row in tile (a % 8)", "dy col in tile (b % 16)", "channel % 64 // 4", "pixels per bad (image, tile)")}
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
        a, b, ch = idx[:, 2] // 2, idx[:, 3] // 2, idx[:, 1]
        hist["dy row in tile (a % 8)"].update((a % 8).tolist())
        hist["dy col in tile (b % 16)"].update((b % 16).tolist())
        hist["channel % 64 // 4"].update(((ch % 64) // 4).tolist())
        tiles = collections.Counter(zip(idx[:, 0].tolist(), (a // 8).tolist(), (b // 16).tolist(), (ch // 64).tolist()))
        hist["pixels per bad (image, tile)"].update(tiles.values())
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
for k, h in hist.items():
    print(f"{k}: {sorted(h.items())}")
print(f"{reps} runs, concurrent load {load}: {tot} bad elements in total")
