"""-m gpu: element-wise parity of the conv kernels THAT THE BENCH RUNS against a plain fp32 torch reference on the CPU.

tests/test_gpu_kernels.py covers every leaf on small maps, where the launcher always picks the narrowest tile.
Here the variants that carry the measured step are selected on purpose:
  * forced (yolo_conv_tune_set / yolo_wgrad_tune_set) on maps whose pixel count is not a multiple of any tile size, so
    the last tile is partial: every channel-tile width of the gather kernel in its register-staged and LDS-DMA forms,
    accumulate on and off, every halo-kernel variant, every weight-gradient tile;
  * by the launcher's own choice on the shapes, row strides and batch size of BASELINE config 2 (preset s, 640x640,
    32 images): the calls of one real training step are recorded and each distinct one is replayed on seeded data --
    forward and data gradient checked on the images where tiles begin, straddle and end, the weight gradient on all 32.
Inputs are rounded to bf16 first, the reference accumulates them in fp32 (F.conv2d / conv2d_input / conv2d_weight),
the kernels accumulate in fp32 and round once: per element |err| <= 2^-8 |ref| + 1e-3 max|ref| (one bf16 ulp of the
element plus accumulation-order noise); fp32 weight gradients 3e-4 of the tensor's max."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def ops():
    from src.hipops import ops as o
    return o


def lib():
    from src.hipops import lib as l
    return l


@pytest.fixture(autouse=True)
def _reset_tuning():
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    yield
    lib().call("yolo_conv_tune_set", 0, -1, -1, -1, -1, 0, 0, 0)
    lib().call("yolo_wgrad_tune_set", 0, 0, 0, 0)
    lib().call("yolo_wgrad_tune_pf", 0)
    lib().call("yolo_conv_wide_set", 2)


def rnd(shape, seed, scale=1.0, dtype=None):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype or BF)


def on_dev(t, ld=None, fill=3.0):
    """NCHW-shaped CPU tensor -> NHWC device tensor; with ld > C a channel slice in the middle of a wider buffer whose
    other channels hold `fill` (a kernel that strays outside its slice reads or clobbers them)."""
    o = ops()
    n, c, h, w = t.shape
    ld = c if ld is None else ld
    buf = o.new_nhwc(n, ld, h, w, t.dtype, DEV)
    buf.fill_(fill)
    off = ((ld - c) // 2) // 8 * 8
    view = buf[:, off:off + c]
    view.copy_(t.to(DEV))
    return view, buf, off


def assert_elem(got, want, what, rel=2.0 ** -8, floor=1e-3):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    lim = rel * want.abs() + floor * float(want.abs().max())
    bad = (got - want).abs() > lim
    if bool(bad.any()):
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements off; first at {idx}: got "
                             f"{float(got[tuple(idx)]):.5f} want {float(want[tuple(idx)]):.5f} (max |ref| {float(want.abs().max()):.3f})")


def assert_slice_untouched(buf, off, c, fill, what):
    """channels of the wider buffer outside [off, off+c) still hold the fill value"""
    outside = torch.cat([buf[:, :off], buf[:, off + c:]], 1)
    assert bool((outside.float() == fill).all()), f"{what}: wrote outside its channel slice"


def run_fwd_dgrad_case(n, cin, cout, h, w, k, s, images, ldx=None, ldy=None, stats=True, acc=(False, True), seed=0,
                       want_plan=None, dtype=None):
    """forward (+ BatchNorm statistics epilogue) and data gradient (+ accumulate form) of one shape against the CPU
    reference evaluated on `images` (conv is per image)."""
    o = ops()
    q = lib().query
    BF = dtype or globals()["BF"]                          # the case's 16-bit type (bf16 unless the test asks for f16)
    rel = 2.0 ** -8 if BF == torch.bfloat16 else 2.0 ** -10
    oh, ow = o.conv_out_hw(h, w, k, s)
    if want_plan is not None:
        got = q("yolo_conv2d_plan", n, h, w, cin, oh, ow, cout, k, s, 0, 0, lib().BF16)
        assert got == want_plan, f"test does not reach the variant it is written for: plan {got}, wanted {want_plan}"
    x, wt = rnd((n, cin, h, w), seed + 1, dtype=BF), rnd((cout, cin, k, k), seed + 2, (cin * k * k) ** -0.5, dtype=BF).float()
    images = sorted(set(i for i in images if i < n))
    wp, wb = o.pack_weights(wt.to(DEV), k, s, 0, BF), o.pack_weights(wt.to(DEV), k, s, 1, BF)
    wref = wt.to(BF).float()
    # ---- forward
    xd, _, _ = on_dev(x, ldx)
    ybuf = o.new_nhwc(n, ldy or cout, oh, ow, BF, DEV).fill_(5.0)
    yoff = (((ldy or cout) - cout) // 2) // 8 * 8
    yv = ybuf[:, yoff:yoff + cout]
    acc_s = o.bn_acc_new(cout, DEV) if stats else None
    y = o.conv_fwd(xd, wp, None, cout, k, s, acc_s, out=yv)
    y_ref = F.conv2d(x[images].float(), wref, None, s, k // 2)
    try:
        assert_elem(y[images], y_ref, f"conv_fwd {(n, cin, h, w, cout, k, s)}", rel=rel)
    except AssertionError as e:
        # say whether the mismatch is a property of the kernel or of this one launch (one flake on record: 3 of 910 200
        # elements two ulps off in a run whose 36 000 repeats were exact -- tools/dbg_wide.py)
        first = y.clone()
        again = o.conv_fwd(xd, wp, None, cout, k, s, o.bn_acc_new(cout, DEV) if stats else None, out=yv)
        same = bool(torch.equal(first, again))
        raise AssertionError(f"{e}; a second launch gives {'the same tensor' if same else 'a DIFFERENT tensor'}: "
                             f"{int((first != again).sum())} elements differ") from None
    assert_slice_untouched(ybuf, yoff, cout, 5.0, "conv_fwd")
    if stats:
        yf = y.float()
        sums = acc_s.view(o.BN_REPL, 2, cout).sum(0).cpu()
        want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]).cpu()
        mass = torch.stack([yf.abs().sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]).cpu()
        assert bool(((sums - want).abs() <= 2e-5 * mass + 1e-3).all()), f"BN statistics epilogue: {(sums - want).abs().max()}"
    # ---- data gradient
    if cin % 8:
        return
    dy = rnd((n, cout, oh, ow), seed + 3, dtype=BF)
    dyd, _, _ = on_dev(dy, ldy)
    dx_ref = torch.nn.grad.conv2d_input((len(images), cin, h, w), wref, dy[images].float(), s, k // 2)
    for accumulate in acc:
        base = rnd((n, cin, h, w), seed + 4, dtype=BF)
        dxv, dxbuf, dxoff = on_dev(base, ldx, fill=9.0)
        if accumulate:
            o.conv_dgrad(dyd, wb, cin, h, w, k, s, acc_into=dxv)
            want = (dx_ref + base[images].float())
            got = dxv
        else:
            got = o.conv_dgrad(dyd, wb, cin, h, w, k, s)
            want = dx_ref
        assert_elem(got[images], want, f"conv_dgrad acc={accumulate} {(n, cin, h, w, cout, k, s)}", rel=rel)
        if accumulate:
            assert_slice_untouched(dxbuf, dxoff, cin, 9.0, "conv_dgrad")


def run_wgrad_case(n, cin, cout, h, w, k, s, ldx=None, ldy=None, seed=0):
    o = ops()
    oh, ow = o.conv_out_hw(h, w, k, s)
    x, dy = rnd((n, cin, h, w), seed + 5), rnd((n, cout, oh, ow), seed + 6)
    xd, _, _ = on_dev(x, ldx)
    dyd, _, _ = on_dev(dy, ldy)
    dw = o.conv_wgrad(xd, dyd, k, s, torch.float32)
    dw_ref = torch.nn.grad.conv2d_weight(x.float(), (cout, cin, k, k), dy.float(), s, k // 2)
    err = float((dw.cpu() - dw_ref).abs().max())
    assert err <= 3e-4 * float(dw_ref.abs().max()), f"conv_wgrad {(n, cin, h, w, cout, k, s)}: {err:.3e} vs max {float(dw_ref.abs().max()):.3e}"


# ------------------------------------------------------------------------------------------ forced variants
# maps of 37 x 41 (1517 pixels per image: no multiple of 128, 16 or 8) so the last pixel tile of every kernel is partial;
# channel counts that are / are not multiples of the channel tile
@pytest.mark.parametrize("bn", [32, 64, 128])
@pytest.mark.parametrize("dma", [0, 1])
@pytest.mark.parametrize("cin,cout,k,s", [(64, 128, 3, 1), (96, 200, 1, 1), (64, 64, 3, 2), (32, 136, 3, 1)])
def test_gather_kernel_every_tile_width_and_staging_mode(bn, dma, cin, cout, k, s):
    lib().call("yolo_conv_tune_set", bn, -1, 0, dma, 0, 0, 0, 0)
    run_fwd_dgrad_case(3, cin, cout, 37, 41, k, s, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16, want_plan=1000 + bn,
                       seed=bn + dma)


@pytest.mark.parametrize("bn", [64, 128])
def test_gather_kernel_tap_inner_order(bn):
    lib().call("yolo_conv_tune_set", bn, 1, 0, 1, 0, 0, 0, 0)
    run_fwd_dgrad_case(2, 64, 128, 23, 29, 3, 1, images=[0, 1], want_plan=1000 + bn, seed=7)


@pytest.mark.parametrize("bk,bm,bn", [(64, 128, 128), (64, 128, 64), (64, 64, 128), (64, 64, 64), (64, 128, 32),
                                      (32, 128, 128), (32, 128, 64), (32, 64, 128), (32, 64, 64)])
@pytest.mark.parametrize("nst", [2, 3, 4])
@pytest.mark.parametrize("cin,cout,k,s", [(64, 128, 3, 1), (96, 200, 1, 1), (64, 64, 3, 2), (160, 136, 3, 1), (128, 72, 3, 2)])
def test_ring_kernel_every_tile_and_depth(bk, bm, bn, nst, cin, cout, k, s):
    """The pipelined ring kernel: every tile shape and K-step at every ring depth; channel counts that are not multiples
    of the 64-deep K-step (96, 160: partial last chunk step) or of the channel tile (200, 136, 72); stride-2 data
    gradients (four parity classes in one launch, odd map: the classes differ in size); partial last pixel tile."""
    lib().call("yolo_conv_tune_set", bn, -1, 0, -1, 1, bm, nst, bk)
    run_fwd_dgrad_case(3, cin, cout, 37, 41, k, s, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16,
                       want_plan=3000 + (500 if bm == 64 else 0) + bn, seed=bm + bn + nst)


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
@pytest.mark.parametrize("cin,cout", [(64, 128), (32, 64), (96, 192)])
def test_halo_kernel_every_variant(variant, cin, cout):
    lib().call("yolo_conv_tune_set", 0, -1, variant, -1, -1, 0, 0, 0)
    run_fwd_dgrad_case(3, cin, cout, 37, 41, 3, 1, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16, want_plan=2000 + variant,
                       seed=variant)


@pytest.mark.parametrize("force", [6, 7, 8, 12])
@pytest.mark.parametrize("cin,cout,h,w", [(64, 128, 23, 20), (32, 64, 13, 40), (96, 200, 9, 20), (128, 72, 20, 40)])
def test_rows_kernel_both_tiles_on_narrow_maps(force, cin, cout, h, w):
    """conv_rows.hip: 20- and 40-pixel-wide maps whose height is no multiple of the 4- / 2-row block, channel counts
    that do not fill the last channel tile"""
    lib().call("yolo_conv_tune_set", 0, -1, force, -1, -1, 0, 0, 0)
    run_fwd_dgrad_case(3, cin, cout, h, w, 3, 1, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16,
                       want_plan=4000 + {6: 1, 7: 2 if cout > 64 else 1, 8: 3, 12: 4}[force], seed=force)


@pytest.mark.parametrize("cin,cout,h,w", [(64, 136, 23, 37), (32, 64, 10, 16), (96, 64, 31, 80)])
def test_rows_kernel_16_pixel_wide_blocks(cin, cout, h, w):
    """conv_rows.hip with 10 x 16-pixel blocks on maps of any size: partial blocks in both directions"""
    lib().call("yolo_conv_tune_set", 0, -1, 14, -1, -1, 0, 0, 0)
    run_fwd_dgrad_case(3, cin, cout, h, w, 3, 1, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16, want_plan=4006, seed=cin)


@pytest.mark.parametrize("cin,cout,h,w", [(16, 32, 23, 37), (32, 16, 41, 16), (64, 24, 20, 33), (16, 16, 7, 50)])
def test_rows_kernel_narrow_layers(cin, cout, h, w):
    """conv_rows.hip, 20 x 16-pixel blocks x 32 channels: fewer than 64 destination channels, 16-channel sources (half a chunk)"""
    lib().call("yolo_conv_tune_set", 0, -1, 16, -1, -1, 0, 0, 0)
    run_fwd_dgrad_case(3, cin, cout, h, w, 3, 1, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16, want_plan=4007, seed=cin + cout)


@pytest.mark.parametrize("force,cin,cout,h,w,plan", [(8, 64, 128, 23, 20, 4003), (12, 96, 72, 13, 40, 4004), (14, 64, 64, 23, 37, 4006)])
def test_rows_kernel_f16(force, cin, cout, h, w, plan):
    """the f16 instantiations of conv_rows.hip (config 5 trains in f16), at f16's tighter tolerance"""
    lib().call("yolo_conv_tune_set", 0, -1, force, -1, -1, 0, 0, 0)
    run_fwd_dgrad_case(3, cin, cout, h, w, 3, 1, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16, want_plan=plan, seed=force,
                       dtype=torch.float16)


def test_stride2_dgrad_patch_kernel_f16():
    q = lib().query
    got = [q("yolo_conv2d_plan", 3, 84, 100, 72, 42, 50, 96, 3, 2, 1, c, lib().F16) for c in range(4)]
    assert all(g == 5008 for g in got), got
    run_fwd_dgrad_case(3, 72, 96, 84, 100, 3, 2, images=[0, 1, 2], ldx=72 + 32, ldy=96 + 16, stats=False, seed=5, dtype=torch.float16)


@pytest.mark.parametrize("pf", [1, 4])          # patches in flight: the big-layer and the small-layer variant of k_wgrad2
@pytest.mark.parametrize("to,ti", [(1, 1), (1, 2), (2, 1), (2, 2)])
@pytest.mark.parametrize("k,s", [(3, 1), (3, 2)])
def test_wgrad_3x3_every_tile(to, ti, k, s, pf):
    lib().call("yolo_wgrad_tune_set", to, ti, 0, 0)
    lib().call("yolo_wgrad_tune_pf", pf)
    assert lib().query("yolo_conv2d_wgrad_plan", 3, 37, 41, 72, *ops().conv_out_hw(37, 41, k, s), 88, k, s, lib().BF16) // 10000000 == pf
    run_wgrad_case(3, 72, 88, 37, 41, k, s, ldx=104, ldy=96, seed=to * 4 + ti)
    if pf == 4:         # a slab of 1, 2, 3, 5 patches: the prefetch ring's prologue / tail shorter than its depth
        for blocks in (4000, 700):
            lib().call("yolo_wgrad_tune_set", to, ti, blocks, 1)
            run_wgrad_case(2, 72, 88, 19, 23, k, s, ldx=104, ldy=96, seed=to * 4 + ti + blocks)


@pytest.mark.parametrize("pf", [1, 4])
@pytest.mark.parametrize("to", [1, 2, 3, 4])
@pytest.mark.parametrize("ti", [1, 2, 3, 4])
def test_wgrad_1x1_every_tile(to, ti, pf):
    lib().call("yolo_wgrad_tune_set", to, ti, 0, 0)
    lib().call("yolo_wgrad_tune_pf", pf)
    run_wgrad_case(3, 136, 120, 37, 41, 1, 1, ldx=160, ldy=128, seed=to * 4 + ti)


# ------------------------------------------------------------------------------------------ the bench's own calls
def _record_step_calls(cfg, n, res):
    """One real bf16 training step of the model; -> distinct dense-conv calls as the launcher saw them."""
    from oracle import blocks as ob
    from oracle.params import det_fill_
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    o = ops()
    calls = {"fwd": set(), "dgrad": set(), "wgrad": set()}
    real = (o.conv_fwd, o.conv_dgrad, o.conv_wgrad)

    def fwd(x, wp, bias, cout, k, stride, stats_acc=None, out=None):
        y = real[0](x, wp, bias, cout, k, stride, stats_acc, out)
        nn, cin, h, w, ldx = o.geom(x)
        calls["fwd"].add((nn, cin, cout, h, w, k, stride, ldx, o.geom(y)[4], stats_acc is not None))
        return y

    def dgrad(dy, wb, cin, h, w, k, stride, acc_into=None, acc2=None):
        dx = real[1](dy, wb, cin, h, w, k, stride, acc_into, acc2)
        nn, cout, _, _, lddy = o.geom(dy)
        calls["dgrad"].add((nn, cin, cout, h, w, k, stride, o.geom(dx)[4], lddy, acc_into is not None))
        return dx

    def wgrad(x, dy, k, stride, w_dtype, out=None):
        nn, cin, h, w, ldx = o.geom(x)
        calls["wgrad"].add((nn, cin, dy.shape[1], h, w, k, stride, ldx, o.geom(dy)[4]))
        return real[2](x, dy, k, stride, w_dtype, out)

    o.conv_fwd, o.conv_dgrad, o.conv_wgrad = fwd, dgrad, wgrad
    try:
        model = Model(**ob.PRESETS[cfg], num_classes=80)
        det_fill_(model.state_dict(), 1)
        model = model.cuda().train()
        g = torch.Generator().manual_seed(3)
        img = torch.randn(n, 3, res, res, generator=g).cuda()
        gts = [torch.tensor([[res * 0.5, res * 0.4, res * 0.2, res * 0.1, 4.]]).cuda() for _ in range(n)]
        with torch.autocast("cuda", dtype=BF):
            preds, a, st = model(img)
            loss, _ = YoloDFLQFLoss(num_classes=80)(preds, gts, a, st)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        o.conv_fwd, o.conv_dgrad, o.conv_wgrad = real
    return calls


@pytest.fixture(scope="module")
def config2_calls():
    return _record_step_calls("s", 32, 640)


def test_config2_forward_and_dgrad_calls_elementwise(config2_calls):
    """Every distinct forward / data-gradient call of preset s @640, 32 images, replayed with the recorded strides and
    the launcher's own variant choice; reference on the images where pixel tiles begin, straddle and end."""
    q = lib().query
    plans, failures = {}, []
    shapes = {}
    for (n, cin, cout, h, w, k, s, ldx, ldy, st) in sorted(config2_calls["fwd"]):
        shapes.setdefault((n, cin, cout, h, w, k, s), [ldx, ldy, st, False])
    for (n, cin, cout, h, w, k, s, lddx, lddy, acc) in sorted(config2_calls["dgrad"]):
        e = shapes.setdefault((n, cin, cout, h, w, k, s), [lddx, lddy, False, acc])
        e[3] = e[3] or acc
    assert len(shapes) >= 20
    for (n, cin, cout, h, w, k, s), (ldx, ldy, st, acc) in shapes.items():
        if cin < 8:
            continue                                     # the 3-channel stem has its own kernels and tests
        oh, ow = ops().conv_out_hw(h, w, k, s)
        p = q("yolo_conv2d_plan", n, h, w, cin, oh, ow, cout, k, s, 0, 0, lib().BF16)
        plans[p] = plans.get(p, 0) + 1
        try:
            run_fwd_dgrad_case(n, cin, cout, h, w, k, s, images=[0, 1, 13, n - 2, n - 1], ldx=ldx, ldy=ldy, stats=st,
                               acc=(False, True) if acc else (False,), seed=cin + cout)
        except AssertionError as e:
            failures.append(str(e))
    print(f"\n[config-2 conv calls] {len(shapes)} distinct shapes, forward plans (kind*1000+width -> count): {plans}")
    assert not failures, "\n".join(failures)
    # the step must have exercised the wide tiles and the halo kernel (what the bench's time is made of)
    assert any(p // 1000 == 3 for p in plans) and any(p // 1000 == 2 for p in plans), plans


def test_config2_wide_and_narrow_epilogue_stores_are_bit_identical_on_every_image(config2_calls):
    """The 16-byte epilogue (store_pixel_blocks: a lane-pair exchange -- by v_permlane16_swap, the default, or by ds_bpermute --
    then one 16-byte store) against the 8-byte form over the WHOLE output of every distinct forward / data-gradient call of
    preset s @640 at 32 images: same values, only the exchange and store instructions differ, so the three must be
    bit-identical on all 32 images, accumulate forms included.  (The element-wise tests compare five images per call against
    the CPU reference.  This all-image test is what found the ring race of round 3 -- a whole workgroup tile of the stride-2
    patch kernel short of one product in images those tests do not look at: DESIGN section 6.)"""
    o, q = ops(), lib().query
    failures, plans = [], {}
    cases = [("fwd",) + c[:9] + (False,) for c in sorted(config2_calls["fwd"])] + [("dgrad",) + c for c in sorted(config2_calls["dgrad"])]
    seen = set()
    for kind, n, cin, cout, h, w, k, s, ld_a, ld_b, acc in cases:
        if cin < 8 or cin % 8 or (kind, n, cin, cout, h, w, k, s, ld_a, ld_b, acc) in seen:
            continue
        seen.add((kind, n, cin, cout, h, w, k, s, ld_a, ld_b, acc))
        oh, ow = o.conv_out_hw(h, w, k, s)
        plan = q("yolo_conv2d_plan", n, h, w, cin, oh, ow, cout, k, s, 1 if kind == "dgrad" else 0, 0, lib().BF16)
        wt = rnd((cout, cin, k, k), 3, (cin * k * k) ** -0.5).float().to(DEV)
        if kind == "fwd":
            src, _, _ = on_dev(rnd((n, cin, h, w), 1), ld_a)
            wp = o.pack_weights(wt, k, s, 0, BF)
            run = lambda out: o.conv_fwd(src, wp, None, cout, k, s, None, out=out)
            shape, ld_out = (n, cout, oh, ow), ld_b
        else:
            src, _, _ = on_dev(rnd((n, cout, oh, ow), 1), ld_b)
            wb = o.pack_weights(wt, k, s, 1, BF)
            run = lambda out: o.conv_dgrad(src, wb, cin, h, w, k, s, acc_into=out if acc else None) if acc else _dgrad_into(o, src, wb, cin, h, w, k, s, out)
            shape, ld_out = (n, cin, h, w), ld_a
        outs = []
        for wide in (2, 1, 0):              # exchange by v_permlane16_swap (default) / by ds_bpermute / 8-byte stores
            lib().call("yolo_conv_wide_set", wide)
            base = rnd(shape, 5)
            dst, _, _ = on_dev(base, ld_out)
            r = run(dst)
            outs.append((r if r is not None else dst).clone())
        lib().call("yolo_conv_wide_set", 2)
        plans[plan // 1000] = plans.get(plan // 1000, 0) + 1
        for name, other in (("permlane vs 8-byte", outs[2]), ("permlane vs bpermute", outs[1])):
            if not torch.equal(outs[0], other):
                bad = (outs[0] != other)
                imgs = bad.flatten(1).any(1).nonzero().flatten().tolist()
                failures.append(f"{kind} {(n, cin, cout, h, w, k, s)} acc={acc} plan {plan} ({name}): {int(bad.sum())} elements differ, images {imgs[:8]}")
    print(f"\n[wide vs narrow stores] {len(seen)} calls, kernel kinds (plan // 1000 -> count): {plans}")
    assert not failures, "\n".join(failures)


@pytest.mark.parametrize("wide", [1, 2, 0])
def test_stride2_patch_kernel_repeats_bit_for_bit_on_every_image(wide):
    """Regression for round 3's ring race (conv_up2.hip: up2_wait): the 128 -> 128 3x3 / 2 data gradient @160x160 at 32 images,
    the launch in which whole workgroup tiles used to come out one product short in a few runs of a hundred (with the
    ds_bpermute epilogue, wide = 1, within a dozen).  Twelve launches, every image, each compared bit for bit with the first;
    the first against the 8-byte-store form."""
    o = ops()
    n, c, h, w = 32, 128, 160, 160
    dy, _, _ = on_dev(rnd((n, c, h // 2, w // 2), 61))
    wt = rnd((c, c, 3, 3), 62, (c * 9) ** -0.5).float().to(DEV)
    wb = o.pack_weights(wt, 3, 2, 1, BF)
    lib().call("yolo_conv_wide_set", 0)
    ref = o.conv_dgrad(dy, wb, c, h, w, 3, 2).clone()
    lib().call("yolo_conv_wide_set", wide)
    side = torch.cuda.Stream()
    a, b = torch.empty(32 << 20, device=DEV), torch.empty(32 << 20, device=DEV)
    for rep in range(12):
        with torch.cuda.stream(side):           # a bandwidth-hungry neighbour, as in the training step
            b.copy_(a)
        got = o.conv_dgrad(dy, wb, c, h, w, 3, 2)
        if not torch.equal(got, ref):
            bad = got != ref
            imgs = bad.flatten(1).any(1).nonzero().flatten().tolist()
            raise AssertionError(f"launch {rep}: {int(bad.sum())} elements differ from the 8-byte form, images {imgs}")
    torch.cuda.synchronize()


def _dgrad_into(o, src, wb, cin, h, w, k, s, out):
    """plain (non-accumulating) data gradient; the result tensor is the kernel's own allocation"""
    return o.conv_dgrad(src, wb, cin, h, w, k, s)


def test_config2_weight_gradient_calls_elementwise(config2_calls):
    """Every distinct weight-gradient call of preset s @640 at the full 32 images (the slab plan depends on the batch)."""
    q = lib().query
    failures, plans = [], set()
    calls = sorted(config2_calls["wgrad"])
    assert len(calls) >= 20
    for (n, cin, cout, h, w, k, s, ldx, ldy) in calls:
        if cin % 8:
            continue
        oh, ow = ops().conv_out_hw(h, w, k, s)
        plans.add(q("yolo_conv2d_wgrad_plan", n, h, w, cin, oh, ow, cout, k, s, lib().BF16))
        try:
            run_wgrad_case(n, cin, cout, h, w, k, s, ldx=ldx, ldy=ldy, seed=cin * 3 + cout)
        except AssertionError as e:
            failures.append(str(e))
    print(f"\n[config-2 wgrad calls] {len(calls)} distinct calls, {len(plans)} distinct plans")
    assert not failures, "\n".join(failures)


@pytest.mark.parametrize("cin,cout,variant", [(32, 64, 16), (72, 96, 8), (128, 32, 8), (40, 64, 8)])
def test_stride2_dgrad_patch_kernel(cin, cout, variant):
    """conv_up2.hip: all four parity classes from one staged dy patch; dy grid 42 x 50 (no multiple of the 8- / 16-row and
    16-column tiles), dx channel counts that do not fill the last 64-channel tile, channel slices of wider buffers,
    accumulate on and off"""
    q = lib().query
    got = [q("yolo_conv2d_plan", 3, 84, 100, cin, 42, 50, cout, 3, 2, 1, c, lib().BF16) for c in range(4)]
    assert all(g == 5000 + variant for g in got), got
    run_fwd_dgrad_case(3, cin, cout, 84, 100, 3, 2, images=[0, 1, 2], ldx=cin + 32, ldy=cout + 16, stats=False, seed=cin + cout)


def test_stride2_dgrad_real_shape_all_parity_classes():
    """128 -> 128 3x3 stride 2 on a 160 x 160 map, 32 images (the biggest conv of preset s): four parity-class launches."""
    q = lib().query
    got = [q("yolo_conv2d_plan", 32, 160, 160, 128, 80, 80, 128, 3, 2, 1, c, lib().BF16) for c in range(4)]
    assert all(g == 5008 for g in got), got                 # conv_up2.hip: the dy patch once for all four classes
    run_fwd_dgrad_case(32, 128, 128, 160, 160, 3, 2, images=[0, 17, 31], stats=True, acc=(False, True), seed=77)
    # the same layer on a 40 x 40 map (dy grid 20 x 20: partial tiles in both directions)
    got = [q("yolo_conv2d_plan", 32, 40, 40, 256, 20, 20, 256, 3, 2, 1, c, lib().BF16) for c in range(4)]
    assert all(g == 5008 for g in got), got
    run_fwd_dgrad_case(32, 256, 256, 40, 40, 3, 2, images=[0, 17, 31], stats=True, acc=(False, True), seed=79)
    # odd map: the parity classes have different sizes
    run_fwd_dgrad_case(2, 64, 64, 45, 39, 3, 2, images=[0, 1], stats=False, seed=78)
