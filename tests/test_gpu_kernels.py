"""-m gpu: every HIP leaf (C-ABI entry point) against its plain-torch stand-in on the same seeded inputs.

Tolerances: fp32 kernels 1e-4 relative to the tensor's max magnitude; bf16/f16 kernels accumulate in fp32
and round once, the stand-in does the same from the same rounded inputs, so 2 output ulps
(bf16: 2^-7, f16: 2^-10 relative) of the tensor's max magnitude.  Integer outputs (pool argmax,
assignment, NMS rows) exact."""
import pytest
import torch

import emulated_ops as emu
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {torch.float32: 1e-4, torch.bfloat16: 2.0 ** -6, torch.float16: 2.0 ** -9}


def ops():
    from src.hipops import ops as o
    return o


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def nhwc(t, pad_c=0):
    """CPU NHWC tensor, optionally as a channel slice of a wider buffer (ld = C + pad_c)."""
    o = ops()
    n, c, h, w = t.shape
    buf = o.new_nhwc(n, c + pad_c, h, w, t.dtype, "cpu")
    if pad_c:
        buf.fill_(7.0)
    view = buf[:, pad_c // 2:pad_c // 2 + c]
    view.copy_(t)
    return view


def dev(t, pad_c=0):
    """Same memory layout on the GPU (keeps the slice/ld structure)."""
    if t is None:
        return None
    if t.dim() != 4 or not ops().is_nhwc(t):
        return t.to(DEV)
    o = ops()
    n, c, h, w = t.shape
    ld = o.geom(t)[4]
    buf = o.new_nhwc(n, ld, h, w, t.dtype, DEV)
    off = (ld - c) // 2 if ld > c else 0
    buf.fill_(7.0)
    view = buf[:, off:off + c]
    view.copy_(t.to(DEV))
    return view


def check(got, want, dtype, what, scale=None, mult=1.0):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    ref = float(want.abs().max()) if scale is None else scale
    err = float((got - want).abs().max())
    lim = TOL[dtype] * mult * max(ref, 1e-6)
    assert err <= lim, f"{what}: max abs err {err:.4e} > {lim:.4e} (ref max {ref:.3e})"


# ------------------------------------------------------------------------------------------ layout / copies
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shapes", [[(9, 7), (5, 4), (3, 2)], [(16, 12), (8, 8), (2, 4)], [(80, 80), (40, 40), (20, 20)]])
def test_head_group_equals_per_branch_pack_and_unpack(dtype, shapes):
    """all six branch tensors <-> (N, no, M) in one launch == one transposing launch per branch; odd maps (2-byte form),
    8-aligned maps incl. the real 80 / 40 / 20 ones (16-byte form, partial last pixel tile), 80 classes"""
    o = ops()
    n = 3
    branches, c_offs, m_offs, m = [], [], [], 0
    for i, (h, w) in enumerate(shapes):
        branches += [dev(nhwc(rnd(n, 64, h, w, seed=60 + i).to(dtype))), dev(nhwc(rnd(n, 80, h, w, seed=70 + i).to(dtype)))]
        c_offs += [0, 64]
        m_offs += [m, m]
        m += h * w
    want = torch.zeros(n, 144, m, dtype=dtype, device=DEV)
    for b, c_off, m_off in zip(branches, c_offs, m_offs):
        o.head_pack(b, want, c_off, m_off)
    got = torch.zeros_like(want)
    o.head_group(branches, got, c_offs, m_offs, True)
    assert torch.equal(got, want)
    outs = [torch.full_like(b, 5.0) for b in branches]
    o.head_group(outs, want, c_offs, m_offs, False)
    for b, x in zip(branches, outs):
        assert torch.equal(b, x)



@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_to_nhwc_and_head_pack_roundtrip(dtype):
    o = ops()
    x = rnd(2, 37, 9, 11, seed=1)
    y = o.to_nhwc(x.to(DEV), dtype)
    assert o.is_nhwc(y) and y.dtype == dtype
    check(y, x.to(dtype), dtype, "to_nhwc")
    preds = torch.full((2, 50, 150), -3.0, dtype=dtype, device=DEV)
    o.head_pack(y, preds, 5, 20)
    want = torch.full((2, 50, 150), -3.0).to(dtype)
    emu.head_pack(x.to(dtype), want, 5, 20)
    assert torch.equal(preds.cpu(), want)
    back = o.head_unpack(preds, 5, 37, 20, 9, 11)
    assert torch.equal(back.cpu(), x.to(dtype))


@pytest.mark.parametrize("dtype,c,pad", [(torch.float32, 12, 4), (torch.bfloat16, 16, 16), (torch.bfloat16, 6, 3)])
def test_copy_channels(dtype, c, pad):
    o = ops()
    src, dst = nhwc(rnd(2, c, 5, 7, seed=2).to(dtype), pad), nhwc(rnd(2, c, 5, 7, seed=3).to(dtype), pad)
    for acc in (False, True):
        d_gpu, d_cpu = dev(dst), dst.clone()
        o.copy_channels(dev(src), d_gpu, acc)
        emu.copy_channels(src, d_cpu, acc)
        check(d_gpu, d_cpu, dtype, f"copy_channels acc={acc}")


@pytest.mark.parametrize("dtype,c,pad,k", [(torch.float32, 12, 4, 2), (torch.bfloat16, 16, 16, 3), (torch.bfloat16, 24, 8, 4),
                                           (torch.float16, 6, 3, 2)])
def test_add_n_gradient_fan_in(dtype, c, pad, k):
    """sum of k channel-slice tensors in one pass (fp32 accumulate, one rounding) == the stand-in's fp32 sum"""
    o = ops()
    xs = [nhwc(rnd(2, c, 5, 7, seed=4 + i).to(dtype), pad if i % 2 else 0) for i in range(k)]
    got = o.add_n([dev(x) for x in xs])
    check(got, emu.add_n(xs), dtype, f"add_n k={k}")
    into = dev(nhwc(rnd(2, c, 5, 7, seed=99).to(dtype), pad))
    o.add_n([dev(x) for x in xs], out=into)
    check(into, emu.add_n(xs), dtype, f"add_n k={k} into a slice")


# ------------------------------------------------------------------------------------------ convolution
CONV_CASES = [  # cin, cout, h, w, k, s, pad_c
    (3, 16, 17, 19, 3, 2, 0), (16, 32, 12, 12, 1, 1, 0), (16, 16, 13, 11, 3, 1, 16), (32, 64, 16, 16, 3, 2, 0),
    (64, 128, 9, 9, 3, 1, 0), (96, 24, 10, 10, 1, 1, 32), (8, 8, 6, 6, 3, 1, 0), (48, 40, 7, 9, 3, 2, 0),
    (128, 256, 8, 8, 1, 1, 0),
]
CONV_MODES = [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1), (torch.float16, 0)]


@pytest.mark.parametrize("dtype,algo", CONV_MODES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype, algo, monkeypatch):
    o = ops()
    monkeypatch.setattr(o, "ALGO", algo)
    cin, cout, h, w, k, s, pad_c = case
    n = 3
    x = nhwc(rnd(n, cin, h, w, seed=10).to(dtype), pad_c)
    wt = rnd(cout, cin, k, k, seed=11, scale=(cin * k * k) ** -0.5)
    bias = rnd(cout, seed=12)
    # forward (+bias)
    y_ref = emu.conv_fwd(x, emu.pack_weights(wt, k, s, 0, dtype), bias, cout, k, s)
    y = o.conv_fwd(dev(x), o.pack_weights(wt.to(DEV), k, s, 0, dtype), bias.to(DEV), cout, k, s)
    check(y, y_ref, dtype, "conv_fwd")
    # dgrad
    oh, ow = y_ref.shape[2:]
    dy = nhwc(rnd(n, cout, oh, ow, seed=13).to(dtype), pad_c)
    if cin >= 8:
        dx_ref = emu.conv_dgrad(dy, emu.pack_weights(wt, k, s, 1, dtype), cin, h, w, k, s)
        dx = o.conv_dgrad(dev(dy), o.pack_weights(wt.to(DEV), k, s, 1, dtype), cin, h, w, k, s)
        check(dx, dx_ref, dtype, "conv_dgrad")
    # wgrad (fp32 accumulation over n*oh*ow products; atomics => order noise)
    dw_ref = emu.conv_wgrad(x, dy, k, s, torch.float32)
    dw = o.conv_wgrad(dev(x), dev(dy), k, s, torch.float32)
    check(dw, dw_ref, torch.float32, "conv_wgrad", mult=4.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("h,w", [(32, 32), (17, 23)])
def test_stem_unfold_path(dtype, h, w):
    o = ops()
    img = rnd(3, 3, h, w, seed=15)
    wt = rnd(16, 3, 3, 3, seed=16, scale=0.2)
    col = o.stem_im2col(img.to(DEV), dtype)
    col_ref = emu.stem_im2col(img, dtype)
    assert torch.equal(col.cpu(), col_ref)
    y = o.conv_fwd(col, o.stem_pack_weights(wt.to(DEV), dtype), None, 16, 1, 1)
    y_ref = emu.conv_fwd(nhwc(torch.as_tensor(img).to(dtype)), emu.pack_weights(wt, 3, 2, 0, dtype), None, 16, 3, 2)
    check(y, y_ref, dtype, "stem forward == 3x3/2 conv")
    dy = nhwc(rnd(*y_ref.shape, seed=17).to(dtype))
    dw = o.stem_unpack_wgrad(o.conv_wgrad(col, dev(dy), 1, 1, torch.float32), torch.float32)
    dw_ref = emu.conv_wgrad(nhwc(img.to(dtype)), dy, 3, 2, torch.float32)
    check(dw, dw_ref, torch.float32, "stem wgrad", mult=4.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cout,h,w", [(16, 32, 32), (32, 17, 23), (64, 640, 322), (48, 9, 515), (96, 6, 6), (128, 34, 258)])
def test_stem_fused_conv(dtype, cout, h, w):
    """conv straight from the NCHW fp32 image == the unfold + 1x1 path (same rounded operands, fp32 accumulation),
    odd sizes, rows longer than one 128-pixel segment, every channel-tile count; statistics of the stored values"""
    o = ops()
    img = rnd(2, 3, h, w, seed=18)
    wt = rnd(cout, 3, 3, 3, seed=19, scale=0.2)
    assert o.stem_conv_eligible(img.to(DEV), dtype, cout)
    wp = o.stem_pack_weights(wt.to(DEV), dtype)
    acc = o.bn_acc_new(cout, DEV)
    y = o.stem_conv_fwd(img.to(DEV), wp, cout, dtype, acc)
    col = o.stem_im2col(img.to(DEV), dtype)
    acc2 = o.bn_acc_new(cout, DEV)
    y2 = o.conv_fwd(col, wp, None, cout, 1, 1, acc2)
    y_ref = emu.conv_fwd(nhwc(img.to(dtype)), emu.pack_weights(wt, 3, 2, 0, dtype), None, cout, 3, 2)
    check(y, y_ref, dtype, "fused stem == 3x3/2 conv")
    assert torch.equal(y.cpu(), y2.cpu()), "fused stem differs from the unfold path"
    st = acc.view(o.BN_REPL, 2, cout).sum(0).cpu()
    yf = y.float().cpu()
    want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    assert torch.allclose(st, want, rtol=2e-4, atol=1e-3 * float(want.abs().max())), (st - want).abs().max()


@pytest.mark.parametrize("dtype,n,cin,cout,h,w,k", [(torch.bfloat16, 2, 64, 64, 20, 20, 3), (torch.bfloat16, 2, 64, 64, 80, 80, 3),
                                                    (torch.bfloat16, 32, 128, 256, 20, 20, 1), (torch.bfloat16, 3, 24, 40, 9, 11, 3),
                                                    (torch.float32, 2, 16, 24, 12, 10, 3), (torch.float16, 2, 128, 64, 40, 40, 3)])
def test_conv_dgrad_with_two_accumulate_sources(dtype, n, cin, cout, h, w, k):
    """dx = dgrad + dx + acc2 in one epilogue == the three terms summed in fp32 and rounded once; the shapes reach the halo,
    ring, gather and generic kernels; dx and acc2 are channel slices of wider buffers (as in C3K2's concat gradient)"""
    o = ops()
    dy = dev(nhwc(rnd(n, cout, h, w, seed=80).to(dtype)))
    wt = rnd(cout, cin, k, k, seed=81, scale=0.1)
    wb = o.pack_weights(wt.to(DEV), k, 1, 1, dtype)
    dx0, a2 = nhwc(rnd(n, cin, h, w, seed=82).to(dtype), 16), nhwc(rnd(n, cin, h, w, seed=83).to(dtype), 32)
    plain = o.conv_dgrad(dy, wb, cin, h, w, k, 1).float().cpu()
    want = (plain + dx0.float() + a2.float()).to(dtype)
    got = o.conv_dgrad(dy, wb, cin, h, w, k, 1, acc_into=dev(dx0), acc2=dev(a2))
    check(got, want, dtype, "dgrad + dst + acc2", mult=2.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cout,h,w,pad", [(16, 32, 32, 0), (32, 17, 23, 16), (64, 640, 322, 0), (48, 9, 515, 0), (96, 6, 6, 16), (128, 34, 258, 0)])
def test_stem_fused_wgrad(dtype, cout, h, w, pad):
    """weight gradient straight from the NCHW fp32 image == the unfold + 1x1 weight gradient (same rounded operands, fp32
    accumulation in another order) and == the fp32 reference of the 3x3/2 conv; odd sizes, partial segments and row pairs,
    every channel-tile count, dy as a channel slice of a wider buffer"""
    o = ops()
    img = rnd(2, 3, h, w, seed=24)
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dy = nhwc(rnd(2, cout, oh, ow, seed=25).to(dtype), pad)
    dw = o.stem_wgrad(img.to(DEV), dev(dy), torch.float32)
    col = o.stem_im2col(img.to(DEV), dtype)
    dw_unf = o.stem_unpack_wgrad(o.conv_wgrad(col, dev(dy), 1, 1, torch.float32), torch.float32)
    scale = float(dw_unf.abs().max())
    assert float((dw - dw_unf).abs().max()) <= 2e-4 * scale, ((dw - dw_unf).abs().max(), scale)
    dw_ref = emu.conv_wgrad(nhwc(img.to(dtype)), dy, 3, 2, torch.float32)
    check(dw, dw_ref, torch.float32, "fused stem wgrad", mult=4.0)
    dwb = o.stem_wgrad(img.to(DEV), dev(dy), torch.bfloat16)
    assert dwb.dtype == torch.bfloat16 and torch.equal(dwb.float().cpu(), dw.cpu().to(torch.bfloat16).float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c,h,w", [(16, 12, 12), (24, 7, 9), (128, 10, 10), (80, 13, 21), (512, 20, 20), (12, 5, 6), (2056, 3, 5)])
def test_depthwise(dtype, c, h, w):
    """odd maps (partial row strips and column blocks), channel groups that do not divide a workgroup (24, 80), more
    channel groups than a workgroup holds (2056), a channel count without 8-alignment (12: the generic kernel)"""
    o = ops()
    x, dy = nhwc(rnd(2, c, h, w, seed=20).to(dtype)), nhwc(rnd(2, c, h, w, seed=21).to(dtype), 16)   # dy: slice of a wider buffer
    w9 = rnd(c, 9, seed=22, scale=0.3)
    check(o.dw_fwd(dev(x), w9.to(DEV)), emu.dw_fwd(x, w9), dtype, "dw_fwd")
    acc = o.bn_acc_new(c, DEV)                                   # forward with the BatchNorm statistics in its epilogue
    ys = o.dw_fwd(dev(x), w9.to(DEV), acc)
    assert torch.equal(ys.cpu(), o.dw_fwd(dev(x), w9.to(DEV)).cpu())
    st, yf = acc.view(o.BN_REPL, 2, c).sum(0).cpu(), ys.float().cpu()
    want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    assert torch.allclose(st, want, rtol=2e-4, atol=1e-3 * float(want.abs().max())), (st - want).abs().max()
    check(o.dw_dgrad(dev(dy), w9.to(DEV)), emu.dw_dgrad(dy, w9), dtype, "dw_dgrad")
    prev = nhwc(rnd(2, c, h, w, seed=23).to(dtype))
    got = o.dw_dgrad(dev(dy), w9.to(DEV), acc_into=dev(prev).clone(memory_format=torch.preserve_format))
    check(got, (emu.dw_dgrad(dy, w9).float() + prev.float()).to(dtype), dtype, "dw_dgrad accumulate", mult=2.0)
    check(o.dw_wgrad(dev(x), dev(dy)), emu.dw_wgrad(x, dy), torch.float32, "dw_wgrad", mult=4.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c,h,w", [(16, 12, 12), (80, 13, 21), (512, 20, 20), (2056, 3, 5)])
def test_fused_inference_depthwise_bias_silu(dtype, c, h, w):
    """yolo_dwconv3x3_fwd_act: act(depthwise(x) + bias) in the strip kernel's epilogue == depthwise, then bias + SiLU in fp32
    on the unrounded sums; a channel count the strip kernel does not take (12) reports 'not taken'."""
    o = ops()
    x = nhwc(rnd(2, c, h, w, seed=24).to(dtype))
    w9, bias = rnd(c, 9, seed=25, scale=0.3), rnd(c, seed=26, scale=0.5)
    ref = torch.nn.functional.conv2d(x.float(), w9.view(c, 1, 3, 3), bias, 1, 1, groups=c)
    for act in (1, 0):
        got = o.dw_fwd_act(dev(x), w9.to(DEV), bias.to(DEV), act)
        assert got is not None
        check(got, torch.nn.functional.silu(ref) if act else ref, dtype, f"dw_fwd_act act={act}", mult=2.0)
    x12 = nhwc(rnd(2, 12, 5, 6, seed=27).to(dtype))
    assert o.dw_fwd_act(dev(x12), rnd(12, 9, seed=28).to(DEV), rnd(12, seed=29).to(DEV), 1) is None


# ------------------------------------------------------------------------------------------ BN + act
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c,act,pad", [(16, 1, 0), (24, 0, 8), (96, 1, 0), (6, 1, 0)])
def test_bn_act_train_fwd_bwd(dtype, c, act, pad):
    o = ops()
    n, h, w = 3, 11, 13
    y = nhwc((rnd(n, c, h, w, seed=30) * 1.5 + 0.3).to(dtype), pad)
    res = nhwc(rnd(n, c, h, w, seed=31).to(dtype))
    gamma, beta = 1 + 0.1 * rnd(c, seed=32), 0.1 * rnd(c, seed=33)
    rm, rv = 0.1 * rnd(c, seed=34), 1 + 0.1 * rnd(c, seed=35).abs()
    rm_g, rv_g = rm.clone().to(DEV), rv.clone().to(DEV)
    ref = emu.bn_train_stats(y, gamma, beta, rm, rv, 0.03, 1e-3)
    got = o.bn_train_stats(dev(y), gamma.to(DEV), beta.to(DEV), rm_g, rv_g, 0.03, 1e-3)
    for g, r, nm in zip(got, ref, ("mean", "invstd", "scale", "shift")):
        check(g, r, torch.float32, nm, mult=2.0)
    check(rm_g, rm, torch.float32, "running_mean"), check(rv_g, rv, torch.float32, "running_var")
    out_ref = emu.bn_act_fwd(y, ref[2], ref[3], act, res)
    out = o.bn_act_fwd(dev(y), got[2], got[3], act, dev(res))
    check(out, out_ref, dtype, "bn_act_fwd")
    dout = nhwc(rnd(n, c, h, w, seed=36).to(dtype), pad)
    dy_ref, dg_ref, db_ref = emu.bn_act_bwd(dout, y, ref[2], ref[3], ref[0], ref[1], gamma, act)
    dy, dg, db = o.bn_act_bwd(dev(dout), dev(y), got[2], got[3], got[0], got[1], gamma.to(DEV), act)
    check(dy, dy_ref, dtype, "bn_bwd dy", mult=2.0)
    check(dg, dg_ref, torch.float32, "dgamma", mult=10.0), check(db, db_ref, torch.float32, "dbeta", mult=10.0)
    check(o.bn_act_bwd_eval(dev(dout), dev(y), got[2], got[3], act), emu.bn_act_bwd_eval(dout, y, ref[2], ref[3], act),
          dtype, "bn_bwd_eval")
    check(o.channel_sum(dev(dout)), emu.channel_sum(dout), torch.float32, "channel_sum", mult=10.0)
    sc, sh = o.bn_eval_coeffs(gamma.to(DEV), beta.to(DEV), rm_g, rv_g, 1e-3)
    sc_r, sh_r = emu.bn_eval_coeffs(gamma, beta, rm, rv, 1e-3)
    check(sc, sc_r, torch.float32, "eval scale"), check(sh, sh_r, torch.float32, "eval shift")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,h,w,k,s,act", [(16, 32, 20, 20, 3, 1, 1), (32, 128, 17, 13, 1, 1, 0), (64, 24, 16, 16, 3, 2, 1)])
def test_bn_accumulator_path_with_conv_epilogue_stats(dtype, cin, cout, h, w, k, s, act):
    """conv writes y AND its channel sums; act kernel derives mean/invstd; backward reduce + apply from acc."""
    o = ops()
    n = 4
    x = nhwc(rnd(n, cin, h, w, seed=60).to(dtype))
    wt = rnd(cout, cin, k, k, seed=61, scale=(cin * k * k) ** -0.5)
    gamma, beta = 1 + 0.1 * rnd(cout, seed=62), 0.1 * rnd(cout, seed=63)
    rm, rv = 0.1 * rnd(cout, seed=64), 1 + 0.1 * rnd(cout, seed=65).abs()
    rm_g, rv_g = rm.clone().to(DEV), rv.clone().to(DEV)
    acc_f, acc_b = o.bn_acc_new(cout, DEV), o.bn_acc_new(cout, DEV)
    ref_f, ref_b = torch.zeros(8 * 2 * cout), torch.zeros(8 * 2 * cout)
    y = o.conv_fwd(dev(x), o.pack_weights(wt.to(DEV), k, s, 0, dtype), None, cout, k, s, acc_f)
    y_ref = emu.conv_fwd(x, emu.pack_weights(wt, k, s, 0, dtype), None, cout, k, s, ref_f)
    check(y, y_ref, dtype, "conv y")
    # statistics of the STORED values: compare with sums of the GPU's own y (isolates the epilogue reduction)
    yf = y.float().cpu()
    sums = acc_f.view(8, 2, cout).sum(0).cpu()
    check(sums[0], yf.sum((0, 2, 3)), torch.float32, "epilogue sum", mult=20.0, scale=float(yf.abs().sum((0, 2, 3)).max()))
    check(sums[1], (yf * yf).sum((0, 2, 3)), torch.float32, "epilogue sumsq", mult=20.0)
    fin = o.bn_finalize_acc(acc_f, n * y.shape[2] * y.shape[3], gamma.to(DEV), beta.to(DEV), rm.clone().to(DEV), rv.clone().to(DEV), 0.03, 1e-3)
    fin_r = emu.bn_finalize_acc(ref_f, n * y.shape[2] * y.shape[3], gamma, beta, rm.clone(), rv.clone(), 0.03, 1e-3)
    for a_, b_, nm in zip(fin, fin_r, ("mean", "invstd", "scale", "shift")):
        check(a_, b_, torch.float32, "finalize_acc " + nm, mult=100.0 if dtype != torch.float32 else 4.0, scale=max(1.0, float(b_.abs().max())))
    res = nhwc(rnd(*y_ref.shape, seed=66).to(dtype))
    out, mean, invstd, scale, shift = o.bn_act_fwd_train(y, acc_f, gamma.to(DEV), beta.to(DEV), rm_g, rv_g, 0.03, 1e-3, act, dev(res))
    out_r, mean_r, invstd_r, scale_r, shift_r = emu.bn_act_fwd_train(y_ref, ref_f, gamma, beta, rm, rv, 0.03, 1e-3, act, res)
    check(scale, scale_r, torch.float32, "scale", mult=100.0 if dtype != torch.float32 else 4.0, scale=max(1.0, float(scale_r.abs().max())))
    check(shift, shift_r, torch.float32, "shift", mult=100.0 if dtype != torch.float32 else 4.0, scale=max(1.0, float(shift_r.abs().max())))
    check(mean, mean_r, torch.float32, "mean", mult=50.0 if dtype != torch.float32 else 4.0, scale=float(y_ref.float().abs().max()))
    check(invstd, invstd_r, torch.float32, "invstd", mult=100.0 if dtype != torch.float32 else 4.0)
    check(out, out_r, dtype, "bn_act_fwd_train", mult=2.0)
    check(rm_g, rm, torch.float32, "running_mean", mult=50.0), check(rv_g, rv, torch.float32, "running_var", mult=100.0)
    dout = nhwc(rnd(*y_ref.shape, seed=67).to(dtype))
    dy, dg, db = o.bn_act_bwd_train(dev(dout), y, scale, shift, mean, invstd, gamma.to(DEV), act, acc_b)
    dy_r, dg_r, db_r = emu.bn_act_bwd_train(dout, y.cpu(), scale.cpu(), shift.cpu(), mean.cpu(), invstd.cpu(), gamma, act, ref_b)
    check(dy, dy_r, dtype, "bn bwd dy", mult=2.0)
    check(dg, dg_r, torch.float32, "dgamma", mult=20.0), check(db, db_r, torch.float32, "dbeta", mult=20.0)


# ------------------------------------------------------------------------------------------ pool / upsample
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_maxpool5_and_upsample(dtype):
    o = ops()
    x = nhwc((rnd(2, 24, 9, 10, seed=40) * 2).round().div(2).to(dtype))        # many exact ties
    out, idx = o.maxpool5_fwd(dev(x))
    out_ref, idx_ref = emu.maxpool5_fwd(x)
    assert torch.equal(out.cpu(), out_ref)
    dout = nhwc(rnd(2, 24, 9, 10, seed=41).to(dtype))
    check(o.maxpool5_bwd(dev(dout), idx), emu.maxpool5_bwd(dout, idx_ref), dtype, "maxpool5_bwd (tie routing)", mult=2.0)
    up = o.upsample2x_fwd(dev(x))
    assert torch.equal(up.cpu(), emu.upsample2x_fwd(x))
    dup = nhwc(rnd(2, 24, 18, 20, seed=42).to(dtype))
    check(o.upsample2x_bwd(dev(dup)), emu.upsample2x_bwd(dup), dtype, "upsample2x_bwd", mult=2.0)


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("heads,dk,dh,h,w", [(2, 32, 64, 6, 6), (4, 32, 64, 20, 20), (1, 16, 32, 13, 11), (3, 32, 64, 10, 19),
                                             (1, 32, 64, 21, 21), (2, 32, 64, 23, 23)])
def test_attention(dtype, heads, dk, dh, h, w):
    """16-bit with dk 32 / dh 64 and <= 448 tokens: the fused flash-style kernels (36, 190, 400, 441 tokens = every tile-pair
    count, partial last tiles and query blocks); 529 tokens and the 16 / 32 head: the batched-GEMM route; fp32: the VALU route"""
    o = ops()
    n = 2
    qkv = nhwc(rnd(n, heads * (2 * dk + dh), h, w, seed=50).to(dtype))
    scale = dk ** -0.5
    og, vg, stash = o.attn_fwd(dev(qkv), heads, dk, dh, scale)
    o_ref, v_ref, lse_ref = emu.attn_fwd(qkv, heads, dk, dh, scale)
    check(og, o_ref, dtype, "attn o", mult=2.0)
    if dtype == torch.float32:          # fp32 path stashes the row log-sum-exp, 16-bit paths the probabilities
        check(stash.view(torch.float32), lse_ref, torch.float32, "lse", mult=4.0)
    assert torch.equal(vg.cpu(), v_ref)
    d_o, d_v = nhwc(rnd(n, heads * dh, h, w, seed=51).to(dtype)), nhwc(rnd(n, heads * dh, h, w, seed=52).to(dtype))
    dq = o.attn_bwd(dev(qkv), og, dev(d_o), dev(d_v), stash, heads, dk, dh, scale)
    dq_ref = emu.attn_bwd(qkv, o_ref, d_o, d_v, lse_ref, heads, dk, dh, scale)
    check(dq, dq_ref, dtype, "attn dqkv", mult=4.0)
    dq0 = o.attn_bwd(dev(qkv), og, dev(d_o), None, stash, heads, dk, dh, scale)          # no gradient through the re-gathered v
    check(dq0, emu.attn_bwd(qkv, o_ref, d_o, None, lse_ref, heads, dk, dh, scale), dtype, "attn dqkv (d_vp None)", mult=4.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("heads,h,w", [(2, 6, 6), (4, 20, 20), (2, 23, 23), (4, 40, 40), (1, 16, 16), (2, 16, 17), (1, 37, 29)])
def test_attention_forward_without_stash_any_length(dtype, heads, h, w):
    """yolo_attn_fwd_nograd (inference / no-grad): <= 448 tokens the one-image kernel without its log-sum-exp store, beyond
    it the key-blocked online-softmax kernel -- 1600 tokens (preset l @1280: the batched-GEMM route before), exactly one and
    two key blocks (256, 272), 1073 = a partial last block and a partial last query tile; scores with a spread that makes
    the running maximum move between blocks; fp32 is not taken (None)."""
    o = ops()
    n, dk, dh = 2, 32, 64
    qkv = rnd(n, heads * (2 * dk + dh), h, w, seed=53)
    qkv[:, :2 * dk] *= 2.5                            # sharper softmax: per-block maxima differ by several units
    qkv = nhwc(qkv.to(dtype))
    scale = dk ** -0.5
    got = o.attn_fwd_nograd(dev(qkv), heads, dk, dh, scale)
    if dtype == torch.float32:
        assert got is None
        return
    o_ref, v_ref, _ = emu.attn_fwd(qkv, heads, dk, dh, scale)
    check(got[0], o_ref, dtype, "attn o (no stash)", mult=2.0)
    assert torch.equal(got[1].cpu(), v_ref)
    if h * w <= 448:                                  # the same kernel as the stashing forward: identical output
        assert torch.equal(got[0], o.attn_fwd(dev(qkv), heads, dk, dh, scale)[0])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,h,w,k,s", [(64, 128, 80, 80, 1, 1),      # k_conv_mfma (large-map 1x1)
                                             (256, 128, 20, 20, 1, 1),     # k_conv_ring (small-map 1x1)
                                             (64, 64, 40, 40, 3, 1),       # k_conv_rows (3x3 on a 40-wide map)
                                             (128, 128, 80, 80, 3, 1),     # k_conv_halo (3x3, two 64-channel tiles)
                                             (32, 64, 37, 41, 3, 2),       # stride 2, partial tiles
                                             (72, 40, 13, 11, 1, 1)])      # channel counts that do not fill a tile
def test_fused_inference_conv_bias_silu_residual(dtype, cin, cout, h, w, k, s):
    """yolo_conv2d_fwd_act: act(conv(x) + bias) (+ residual) in the conv's own epilogue (Model.fuse() blocks) == conv, then the
    element-wise pass, for SiLU / identity, with / without a residual that is a channel slice of a wider buffer."""
    o = ops()
    n = 2
    x = nhwc(rnd(n, cin, h, w, seed=80).to(dtype))
    wt = rnd(cout, cin, k, k, seed=81, scale=(cin * k * k) ** -0.5)
    bias = rnd(cout, seed=82, scale=0.5)
    oh, ow = o.conv_out_hw(h, w, k, s)
    res = nhwc(rnd(n, cout, oh, ow, seed=83).to(dtype), 16)
    wp = o.pack_weights(wt.to(DEV), k, s, 0, dtype)
    ref_y = torch.nn.functional.conv2d(x.float(), wt.to(dtype).float(), bias, s, k // 2)
    for act in (1, 0):
        for r in (None, res):
            got = o.conv_fwd_act(dev(x), wp, bias.to(DEV), cout, k, s, act, None if r is None else dev(r))
            assert got is not None, "no MFMA kernel took this shape"
            want = torch.nn.functional.silu(ref_y) if act else ref_y
            if r is not None:
                want = want + r.float()
            check(got, want, dtype, f"conv_fwd_act act={act} res={r is not None}", mult=2.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cout,h,w", [(64, 1280, 1280), (32, 17, 23), (48, 9, 515)])
def test_fused_inference_stem_bias_silu(dtype, cout, h, w):
    """yolo_stem_conv_fwd with bias + act: the fused model's first block (3 -> C, 3x3 / 2; no MFMA conv kernel takes Cin = 3,
    without this epilogue it ran on the scalar kernel: 4.9 of 10.6 ms of preset l @1280) == SiLU(conv(img) + b); with
    statistics requested the combination is refused."""
    o = ops()
    n = 1 if h > 1000 else 2
    img = rnd(n, 3, h, w, seed=84)
    wt = rnd(cout, 3, 3, 3, seed=85, scale=0.2)
    bias = rnd(cout, seed=86, scale=0.5)
    wp = o.stem_pack_weights(wt.to(DEV), dtype)
    ref = torch.nn.functional.conv2d(img.to(dtype).float(), wt.to(dtype).float(), bias, 2, 1)
    for act in (1, 0):
        got = o.stem_conv_fwd(img.to(DEV), wp, cout, dtype, None, bias.to(DEV), act)
        check(got, torch.nn.functional.silu(ref) if act else ref, dtype, f"fused stem act={act}", mult=2.0)
    with pytest.raises(Exception):
        o.stem_conv_fwd(img.to(DEV), wp, cout, dtype, o.bn_acc_new(cout, DEV), bias.to(DEV), 1)


# ------------------------------------------------------------------------------------------ loss
def _pack(gts):
    from src.model.losses import PackedTargets
    return PackedTargets([g.to(DEV) for g in gts], DEV)


def _loss_case(preds, gts, anchors, strides, nc, dtype, lam=(1.5, 1.0)):
    o = ops()
    pk = _pack(gts)
    p = preds.to(dtype)
    out, dp, ws = o.loss_fwd_bwd(p.to(DEV), anchors.to(dtype).to(DEV), strides.to(dtype).to(DEV), *pk.as_tuple(), nc,
                                 lam[0], lam[1], True)
    ref_out, ref_dp, _ = emu.loss_fwd_bwd(p, anchors, strides, pk.gt.cpu(), pk.gt_off.cpu(), pk.gt_img.cpu(), pk.n_gt, nc,
                                          lam[0], lam[1], True)
    n, _, a = preds.shape
    idx = ws[n * a * 16:n * a * 16 + 4 * max(pk.n_gt, 1)].view(torch.int32)[:pk.n_gt].cpu()
    return out.cpu(), dp.float().cpu(), ref_out, ref_dp.float(), idx


def test_loss_golden_small_fp32():
    gd = load_golden("loss_small")
    gts = [gd["gt0"], gd["gt1"].reshape(0, 5), gd["gt2"]]
    out, dp, ref_out, ref_dp, _ = _loss_case(gd["preds"], gts, gd["anchors"], gd["strides"], 8, torch.float32)
    want = torch.tensor([float(gd["total"]), float(gd["box"]), float(gd["cls"])])
    assert torch.allclose(out, want, rtol=1e-5, atol=1e-6), (out, want)           # vs the reference itself
    assert torch.allclose(dp, gd["dpreds"], rtol=1e-4, atol=1e-7), float((dp - gd["dpreds"]).abs().max())


def test_loss_golden_n320_fp32():
    from test_oracle_golden import n320_inputs
    gd, preds, gts, a, s = n320_inputs()
    out, dp, ref_out, ref_dp, _ = _loss_case(preds, gts, a, s, 80, torch.float32)
    want = torch.tensor([float(gd["total"]), float(gd["box"]), float(gd["cls"])])
    assert torch.allclose(out, want, rtol=1e-5, atol=1e-6), (out, want)
    assert torch.allclose(dp[:, :, ::25], gd["dpreds_stride25"], rtol=1e-4, atol=1e-9)
    assert abs(float(dp.double().abs().sum()) - float(gd["dpreds_abs"])) <= 1e-5 * float(gd["dpreds_abs"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_loss_random_s640_shape(dtype):
    """N=4, A=8400, nc=80, 1..20 GTs per image (COCO-shaped synthetic targets)."""
    from oracle.blocks import make_anchors
    g = torch.Generator().manual_seed(70)
    preds = torch.randn(4, 144, 8400, generator=g)
    preds[:, 64:] = preds[:, 64:] * 0.7 - 3.5
    a, s = make_anchors([(80, 80), (40, 40), (20, 20)], [8.0, 16.0, 32.0])
    gts = []
    for i in range(4):
        m = int(torch.randint(1, 21, (1,), generator=g))
        gts.append(torch.cat([torch.rand(m, 2, generator=g) * 640, torch.rand(m, 2, generator=g) * 256 + 8,
                              torch.randint(0, 80, (m, 1), generator=g).float()], 1))
    out, dp, ref_out, ref_dp, idx = _loss_case(preds, gts, a.t().contiguous(), s.t().contiguous(), 80, dtype)
    assert torch.allclose(out, ref_out, rtol=1e-4 if dtype == torch.float32 else 1e-3, atol=1e-6), (out, ref_out)
    lim = 1e-4 if dtype == torch.float32 else 2.0 ** -6
    err = float((dp - ref_dp).abs().max()) / float(ref_dp.abs().max())
    assert err <= lim, err


def test_module_level_loss_helpers_match_the_oracle_with_gradients():
    """bbox_iou / quality_focal_loss / distribution_focal_loss (reference src/model/losses.py:9,46,63) as device ops:
    values and gradients against the oracle's differentiable restatements (fp32, 1e-5)."""
    from oracle import loss as ol
    from src.model.losses import bbox_iou, distribution_focal_loss, quality_focal_loss
    g = torch.Generator().manual_seed(71)
    m = 333
    b1 = torch.cat([torch.rand(m, 2, generator=g) * 100, torch.rand(m, 2, generator=g) * 40 + 5], 1)
    b2 = b1 + torch.randn(m, 4, generator=g) * 6
    b2[:, 2:] = b2[:, 2:].abs() + 1
    b2[:7] = b1[:7] + 500                                   # disjoint boxes: clamp branch
    w = torch.randn(m, generator=g)

    def both(fn_dev, fn_ref, args, weight=None):
        dev_args = [a.clone().cuda().requires_grad_(True) for a in args]
        ref_args = [a.clone().requires_grad_(True) for a in args]
        od, orf = fn_dev(*dev_args), fn_ref(*ref_args)
        assert torch.allclose(od.cpu(), orf, rtol=1e-5, atol=1e-6), float((od.cpu() - orf).abs().max())
        if weight is None:
            od.backward(), orf.backward()
        else:
            od.backward(weight.cuda()), orf.backward(weight)
        for a, b in zip(dev_args, ref_args):
            assert torch.allclose(a.grad.cpu(), b.grad, rtol=2e-4, atol=1e-6), float((a.grad.cpu() - b.grad).abs().max())

    both(bbox_iou, ol.quirk_iou, (b1, b2), w)
    logits = torch.randn(57, 80, generator=g) * 2
    target = torch.rand(57, 80, generator=g) * (torch.rand(57, 80, generator=g) < 0.05)
    both(quality_focal_loss, ol.qfl_sum, (logits, target))
    both(lambda p, t: quality_focal_loss(p, t, beta=1.5), lambda p, t: ol.qfl_sum(p, t, 1.5), (logits, target))
    dist_logits = torch.randn(91, 16, generator=g)
    tv = torch.rand(91, generator=g) * 14.98
    both(distribution_focal_loss, ol.dfl_side, (dist_logits, tv))


# ------------------------------------------------------------------------------------------ decode / NMS
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_head_decode_and_dfl(dtype):
    o = ops()
    gd = load_golden("decode_val")
    p = gd["preds"].to(dtype)
    y = o.head_decode(p.to(DEV), gd["anchors"].to(DEV), gd["strides"].to(DEV), 8)
    check(y, emu.head_decode(p, gd["anchors"], gd["strides"], 8), dtype, "head_decode")
    x = load_golden("block_dfl")["x"].to(dtype)
    check(o.dfl_expect(x.to(DEV)), emu.dfl_expect(x), dtype, "dfl_expect")


def _nms_compare(pred, nc, **kw):
    from src.utils.model_utils import non_max_suppression
    from oracle.postproc import non_max_suppression as ref_nms
    got = non_max_suppression(pred.to(DEV), nc=nc, **kw)
    want = ref_nms(pred.clone(), nc=nc, **kw)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape, (i, g.shape, w.shape)
        assert torch.equal(g.cpu(), w.float()), f"image {i}: rows differ"


@pytest.mark.parametrize("tag,kw", [
    ("default", dict(conf_thres=0.25, iou_thres=0.45)), ("agnostic", dict(conf_thres=0.25, iou_thres=0.45, agnostic=True)),
    ("multi", dict(conf_thres=0.6, iou_thres=0.5, multi_label=True)), ("classes", dict(conf_thres=0.25, iou_thres=0.45, classes=[1, 3, 6])),
    ("maxdet", dict(conf_thres=0.25, iou_thres=0.9, max_det=17)), ("highconf", dict(conf_thres=0.999, iou_thres=0.45))])
def test_nms_golden_rows_exact(tag, kw):
    from src.utils.model_utils import non_max_suppression
    gd = load_golden("nms")
    got = non_max_suppression(gd["prediction"].to(DEV), nc=8, **kw)
    for i in range(2):
        want = gd[f"{tag}:{i}"]
        want = want.reshape(-1, 6) if isinstance(want, torch.Tensor) else torch.zeros(0, 6)
        assert got[i].shape == want.shape, (tag, i, got[i].shape, want.shape)
        assert torch.equal(got[i].cpu(), want), (tag, i)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_nms_stress_config5_shape(dtype):
    """(bs, 84, 33600), thousands of candidates, >= 300 kept: BASELINE config 5's NMS tensor."""
    g = torch.Generator().manual_seed(90)
    m = 33600
    pred = torch.empty(2, 84, m)
    pred[:, 0:2] = torch.rand(2, 2, m, generator=g) * 1280
    pred[:, 2:4] = torch.rand(2, 2, m, generator=g) * 120 + 10
    pred[:, 4:] = torch.rand(2, 80, m, generator=g) * 0.93
    _nms_compare(pred.to(dtype), 80, conf_thres=0.9, iou_thres=0.45)
