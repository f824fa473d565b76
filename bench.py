#!/usr/bin/env python3
"""Headline benchmark: images/sec of full training steps (fwd + DFL/QFL loss + bwd + grad sync + AdamW) of the
`s` preset at 640x640 bf16 on synthetic COCO-shaped batches, 32 images per GPU (BASELINE.json configs[1]/[2]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = all ranks' images / max-over-ranks time of exactly K steps with the
inputs resident in HBM.  `roofline` = the dominant kernel group of the step measured live with stream events
in an instrumented (eager) step: algorithmic FLOPs of those launches / their summed duration vs the dense bf16
MFMA peak.  `cpu_baseline` (rank 0, N=1 only) = the CPU oracle restatement timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "custom-yolo-implmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PRESET_S = dict(csp=[False, True], depth=[1] * 6, width=[3, 32, 64, 128, 256, 512])
PRESETS = {"s": PRESET_S,                                                              # the headline workload
           "n": dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256]),
           "l": dict(csp=[True, True], depth=[2] * 6, width=[3, 64, 128, 256, 512, 512]),
           "x": dict(csp=[True, True], depth=[2] * 6, width=[3, 96, 192, 384, 768, 768])}
MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
HBM_MEASURED_GBS = 6290.0             # stream rate measured on this part, MI355X_MICROARCH.md


def synthetic_batch(n, res, nc, seed, device):
    """randn images; per image 1..20 boxes (cx,cy,w,h,cls) in pixels -- the collate_fn format."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n, 3, res, res, generator=g)
    gts = []
    for _ in range(n):
        m = int(torch.randint(1, 21, (1,), generator=g))
        gts.append(torch.cat([torch.rand(m, 2, generator=g) * res, torch.rand(m, 2, generator=g) * (0.4 * res) + 8,
                              torch.randint(0, nc, (m, 1), generator=g).float()], 1))
    return img.to(device), [t.to(device) for t in gts]


class KernelTimer:
    """Brackets selected leaf ops with events on torch's current stream (the stream the kernels launch on)."""

    def __init__(self, ops):
        self.ops, self.rec, self.saved, self.detail = ops, [], {}, []

    def _wrap(self, name, group, flops_fn, bytes_fn):
        fn = getattr(self.ops, name)
        self.saved[name] = fn

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.rec.append((group, e0, e1, flops_fn(out, *a) if flops_fn else 0.0, bytes_fn(out, *a) if bytes_fn else 0.0))
            if os.environ.get("BENCH_VERBOSE"):
                shapes = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor) and t.dim() == 4]
                self.detail.append((name, shapes, [x for x in a if isinstance(x, int)], len(self.rec) - 1))
            return out
        setattr(self.ops, name, timed)

    def install(self):
        def conv_flops(out, x, wp, bias, cout, k, stride, *_):
            n, _, oh, ow = out.shape
            return 2.0 * n * oh * ow * cout * x.shape[1] * k * k

        def dgrad_flops(out, dy, wb, cin, h, w, k, stride, *_):
            n, cout, oh, ow = dy.shape
            return 2.0 * n * oh * ow * cout * cin * k * k

        def wgrad_flops(out, x, dy, k, stride, wd, *_):
            n, cout, oh, ow = dy.shape
            return 2.0 * n * oh * ow * cout * x.shape[1] * k * k

        def io_bytes(out, *a):
            tot = 0
            outs = (out,) if isinstance(out, torch.Tensor) else tuple(out or ())
            for t in a + outs:
                if isinstance(t, torch.Tensor) and t.dim() == 4:
                    tot += t.numel() * t.element_size()
            return float(tot)

        def bn_bwd_bytes(out, dout, y, *_):          # pass 1 reads dout, y; pass 2 reads them again and writes dz
            return 5.0 * dout.numel() * dout.element_size()

        def bn_fwd_bytes(out, y, scale, shift, act, res=None, *_):
            return (2.0 + (1.0 if res is not None else 0.0)) * y.numel() * y.element_size()

        def bn_fwd_train_bytes(out, y, acc, gamma, beta, rm, rv, momentum, eps, act, res=None, *_):
            return (2.0 + (1.0 if res is not None else 0.0)) * y.numel() * y.element_size()

        CONV = "conv_mfma(fwd+dgrad)"
        self._wrap("conv_fwd", CONV, conv_flops, io_bytes)
        self._wrap("conv_dgrad", CONV, dgrad_flops, io_bytes)
        self._wrap("conv_wgrad", "wgrad_mfma", wgrad_flops, io_bytes)
        # BatchNorm + SiLU (+ residual): what a Model in training mode calls (statistics come from the conv epilogue) ...
        self._wrap("bn_act_fwd_train", "bn_forward(normalize+act)", None, bn_fwd_train_bytes)
        self._wrap("bn_act_bwd_train", "bn_backward(reduce+apply)", None, bn_bwd_bytes)
        # ... and the two-level forms (deterministic mode, blocks outside a Model, eval)
        self._wrap("bn_act_bwd", "bn_backward(reduce+apply)", None, bn_bwd_bytes)
        self._wrap("bn_act_fwd", "bn_forward(normalize+act)", None, bn_fwd_bytes)
        self._wrap("bn_train_stats", "bn_forward(normalize+act)", None, io_bytes)
        for nm in ("dw_fwd", "dw_dgrad", "dw_wgrad"):
            self._wrap(nm, "depthwise3x3", None, io_bytes)
        for nm in ("stem_conv_fwd", "stem_wgrad", "stem_im2col"):
            self._wrap(nm, "stem(image->32ch)", None, io_bytes)
        for nm in ("bn_stats_acc", "bn_finalize_acc", "copy_channels", "maxpool5_fwd", "maxpool5_bwd", "upsample2x_fwd",
                   "upsample2x_bwd", "head_group", "add_n", "channel_sum"):
            self._wrap(nm, "elementwise(copy/pool/upsample/head transposes)", None, io_bytes)
        self._wrap("attn_fwd", "attention", None, io_bytes)
        self._wrap("attn_bwd", "attention", None, io_bytes)
        self._wrap("loss_fwd_bwd", "loss", None, lambda out, preds, *_: 2.0 * preds.numel() * preds.element_size())
        return self

    def remove(self):
        for k, v in self.saved.items():
            setattr(self.ops, k, v)

    def summary(self):
        torch.cuda.synchronize()
        if self.detail:          # BENCH_VERBOSE=1: the 40 longest leaf calls with shapes, TFLOP/s and GB/s
            rows = []
            for name, shapes, ints, i in self.detail:
                _, e0, e1, fl, by = self.rec[i]
                ms = e0.elapsed_time(e1)
                rows.append((ms, name, shapes, ints, fl / ms / 1e9 if ms else 0, by / ms / 1e6 if ms else 0))
            top = len(rows) if os.environ.get("BENCH_VERBOSE") == "2" else 40
            for ms, name, shapes, ints, tf, gbs in sorted(rows, key=lambda r: -r[0])[:top]:
                print(f"[detail] {1e3 * ms:8.1f} us {name:14s} {tf:7.1f} TF/s {gbs:7.0f} GB/s {shapes} {ints}", file=sys.stderr)
        groups = {}
        for grp, e0, e1, fl, by in self.rec:
            g = groups.setdefault(grp, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, ideal_ms=0.0, ideal_ms_hbm_measured=0.0))
            g["ms"] += e0.elapsed_time(e1)
            g["launches"] += 1
            g["flops"] += fl
            g["bytes"] += by
            # the launch's own roof: the larger of its MFMA time and its HBM time (per-layer min(MFMA, HBM) rate)
            g["ideal_ms"] += 1e3 * max(fl / (MFMA_BF16_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9))
            g["ideal_ms_hbm_measured"] += 1e3 * max(fl / (MFMA_BF16_PEAK_TFLOPS * 1e12), by / (HBM_MEASURED_GBS * 1e9))
        return groups


def measure_preset(preset, batch, res, nc, dev, steps, warmup, deterministic=False):
    """The same captured training step on another preset (extra data point, not the metric).  `deterministic`: the
    package's deterministic mode (fixed-order BatchNorm statistics instead of float atomics): the price of bit-reproducibility."""
    from src.hipops import functions as F_
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.fused_adamw import HipAdamW
    from src.training.graph_step import TrainStepRunner
    torch.manual_seed(0)
    model = Model(**PRESETS[preset], num_classes=nc).to(dev).train()
    opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    img, gts = synthetic_batch(batch, res, nc, 4321, dev)
    was, F_.DETERMINISTIC = F_.DETERMINISTIC, deterministic
    try:
        runner = TrainStepRunner(model, YoloDFLQFLoss(num_classes=nc), opt, "bfloat16", use_graph=True)
        runner.capture(img, PackedTargets(gts, dev))
    finally:
        F_.DETERMINISTIC = was
    for _ in range(warmup):
        runner.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = runner.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(images_per_s=round(batch * steps / dt, 1), ms_per_step=round(1e3 * dt / steps, 3), steps=steps, batch=batch,
                final_loss=round(float(loss), 5))


def measure_fsdp2(preset, batch, res, nc, dev, steps, warmup, precision="bfloat16"):
    """BASELINE config 4's path per GPU: the model under `prepare_fsdp2_model` (fully_shard per C3K2 / SPPF / PSA + root,
    bf16 parameters AND BatchNorm buffers, no autocast: reference src/training/utils_train.py:116-165) stepped by the
    reference's loop body (src/training/train_model.py:234-253: zero_grad, forward, loss, backward, optimizer step) on a
    one-rank RCCL group -- every all-gather / reduce-scatter is issued, over one rank.  Eager (FSDP2's hooks are host
    code), so this is the host-bound figure of the sharded path; the unsharded captured step of the same model is
    `preset_l_640_bf16_16img`."""
    import socket
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.utils_train import get_optimizer, prepare_fsdp2_model
    own = not dist.is_initialized()
    # RCCL prints its version banner to STDOUT when the communicator comes up; bench.py's stdout carries exactly one JSON line
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        if own:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    try:
        torch.manual_seed(0)
        model = Model(**PRESETS[preset], num_classes=nc)
        model = prepare_fsdp2_model(model=model, device_id=dev.index, config={"precision": precision}, world_size=1, device="cuda").train()
        opt, _ = get_optimizer(model, lr=1e-4, weight_decay=1e-4, patience=3, factor=0.5)
        crit = YoloDFLQFLoss(num_classes=nc)
        img, gts = synthetic_batch(batch, res, nc, 4321, dev)

        def step():
            opt.zero_grad()
            preds, anchors, strides = model(img)
            loss, ld = crit(preds, gts, anchors, strides)
            loss.backward()
            opt.step()
            return loss
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        # a HOST-bound figure (the device idles: tools/fsdp2_steps.py) on a box whose host cores are shared: three windows, the
        # fastest one reported, all three listed
        windows = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            torch.cuda.synchronize()
            windows.append(time.perf_counter() - t0)
        dt = min(windows)
        groups = sum(1 for m in model.modules() if type(m).__name__.startswith("FSDP"))
        return dict(images_per_s=round(batch * steps / dt, 1), ms_per_step=round(1e3 * dt / steps, 3), steps=steps, batch=batch,
                    windows_ms_per_step=[round(1e3 * w / steps, 2) for w in windows],
                    final_loss=round(float(loss), 5), wrapper="prepare_fsdp2_model", fsdp_groups=groups, world=1, eager=True,
                    optimizer=type(opt).__name__, param_dtype=precision)
    finally:
        if own:
            dist.destroy_process_group()


def measure_sharded(preset, batch, res, nc, dev, steps, warmup, precision="bfloat16"):
    """BASELINE config 4 per GPU through `ShardedStepRunner` (src/training/sharded_step.py): the reference's FSDP2 numeric
    contract (bf16 parameters and BatchNorm buffers, no autocast, fp32 master shards stepped by AdamW) with the step captured
    as forward/backward/pack -> reduce-scatter -> shard update -> all-gather; on a one-rank RCCL group both collectives are
    issued over one rank.  Bit-identical to the torch-FSDP2 step above in deterministic mode (tests/test_gpu_fsdp.py)."""
    import socket
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.sharded_step import ShardedStepRunner
    own = not dist.is_initialized()
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        if own:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    try:
        torch.manual_seed(0)
        model = Model(**PRESETS[preset], num_classes=nc).to(dev).train()
        runner = ShardedStepRunner(model, YoloDFLQFLoss(num_classes=nc), precision=precision, lr=1e-4, weight_decay=1e-4)
        img, gts = synthetic_batch(batch, res, nc, 4321, dev)
        runner.capture(img, PackedTargets(gts, dev))
        for _ in range(warmup):
            runner.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = runner.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dict(images_per_s=round(batch * steps / dt, 1), ms_per_step=round(1e3 * dt / steps, 3), steps=steps, batch=batch,
                    final_loss=round(float(loss), 5), runner="ShardedStepRunner", world=1, hip_graph=runner.graph is not None,
                    collectives_per_step=["reduce_scatter_tensor AVG", "all_gather_into_tensor"],
                    bytes_per_collective=int(runner.flat_g.numel() * runner.flat_g.element_size()), param_dtype=precision,
                    master_shard_elems=int(runner.master.numel()))
    finally:
        if own:
            dist.destroy_process_group()


def config5_nms_tensor(bs=8, nc=80, m=33600, seed=0):
    """BASELINE config 5's NMS / IoU stress tensor: (bs, 4+nc, 33600), boxes random cxcywh on a 1280 canvas,
    ~15 % of the anchors confident (several thousand candidates per image after conf_thres 0.25, >= 300 kept)."""
    g = torch.Generator().manual_seed(seed)
    pred = torch.empty(bs, 4 + nc, m)
    pred[:, 0:2] = torch.rand(bs, 2, m, generator=g) * 1280
    pred[:, 2:4] = torch.rand(bs, 2, m, generator=g) * 200 + 20
    pred[:, 4:] = torch.rand(bs, nc, m, generator=g) * 0.2
    hot = torch.rand(bs, m, generator=g) < 0.15
    cls = torch.randint(0, nc, (bs, m), generator=g)
    val = (torch.rand(bs, 1, m, generator=g) * 0.7 + 0.3) * hot.unsqueeze(1) + 0.1 * (~hot).unsqueeze(1)
    pred[:, 4:].scatter_(1, cls.unsqueeze(1), val)
    return pred


def measure_inference(dev, preset="l", res=1280, batch=4, nc=80, reps=10):
    """BASELINE config 5's model half: `Model.fuse()` + forward + decode + class-aware NMS of preset l at 1280 x 1280 in fp16.
    A random-weight network without batch statistics saturates its fp16 logits, so after the (timed) forward the class rows
    of the prediction are overwritten with the class rows of config 5's NMS tensor and the DFL rows with unit noise (one
    device copy, timed): ~5000 candidates per image pass conf_thres, 300 are kept -- the density `nms_config5` is measured
    on.  ms / image."""
    from src.hipops import ops
    from src.model.model_builder import Model
    from src.utils.model_utils import non_max_suppression
    torch.manual_seed(0)
    model = Model(**PRESETS[preset], num_classes=nc).to(dev).eval().fuse()
    img = torch.randn(batch, 3, res, res, device=dev)
    m = int(sum((res // s) ** 2 for s in (8, 16, 32)))
    synth = torch.cat([torch.randn(batch, 64, m), config5_nms_tensor(bs=batch, nc=nc, m=m, seed=3)[:, 4:]], 1).half().to(dev)
    cand = []

    def run(count=False):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            preds, anchors, strides = model(img)
            preds.copy_(synth)
            y = ops.head_decode(preds, anchors, strides, nc)
            if count:
                cand.append(int((y[:, 4:].amax(1) > 0.25).sum()) // batch)
            return non_max_suppression(y, conf_thres=0.25, iou_thres=0.45, nc=nc)
    out = run(count=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    full = (time.perf_counter() - t0) / reps * 1e3
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model(img)
        torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / reps * 1e3
    return dict(ms_per_image=round(full / batch, 3), forward_ms_per_image=round(fwd / batch, 3), images_per_s=round(1e3 * batch / full, 1),
                batch=batch, anchors=m, candidates_per_image=cand[0],
                kept_per_image=sum(o.shape[0] for o in out) // batch, eager_launches=True,
                forward_tflops=round(347.64 * batch / fwd, 1) if (preset, res) == ("l", 1280) else None)


def measure_single_image(dev, preset, res=640, nc=80, reps=30):
    """`Model.inference(image)` -- the reference's one-image entry point -- on a fused fp16 model: ms per call with the
    forward + decode replayed as one hipGraph (the default, src/model/infer_graph.py) and launch by launch."""
    from src.model.model_builder import Model
    torch.manual_seed(0)
    model = Model(**PRESETS[preset], num_classes=nc).to(dev).eval().fuse()
    img = torch.randn(1, 3, res, res, device=dev)
    out = {}
    was = Model.graph_inference
    try:
        for name, flag in (("replayed_ms", True), ("eager_ms", False)):
            Model.graph_inference = flag
            with torch.autocast("cuda", dtype=torch.float16):
                for _ in range(3):
                    model.inference(img)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    model.inference(img)
                torch.cuda.synchronize()
            out[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    finally:
        Model.graph_inference = was
    out.update(res=res, hip_graph=bool(model._infer_graphs is not None and model._infer_graphs.disabled is None))
    return out


def measure_epoch_loop(dev, nc=80, steps=40):
    """The loop a user runs -- src/training/train_model.py::_run_epoch with the captured step -- over batches that live in
    pinned HOST memory (32 x 3 x 640 x 640 fp32 = 157 MB per step): the PCIe-inclusive rate.  The next batch is uploaded on a
    second stream beside the current step (DevicePrefetcher) and the loss scalars stay on the device."""
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training import train_model as tm
    from src.training.fused_adamw import HipAdamW
    torch.manual_seed(0)
    model = Model(**PRESETS["s"], num_classes=nc).to(dev).train()
    opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    crit = YoloDFLQFLoss(num_classes=nc)
    img, gts = synthetic_batch(32, 640, nc, 1234, dev)
    host = [img.cpu().pin_memory() for _ in range(3)]
    targets = [{"boxes": g.cpu()} for g in gts]

    class Loader(list):
        sampler = None
    cap = tm.CapturedTraining(model, crit, opt, "bfloat16")
    kw = dict(device_type="cuda", dtype=torch.bfloat16, enabled=True)
    tm._run_epoch(model, Loader((host[i % 3], targets) for i in range(6)), crit, "cuda", kw, -1, "warm-up", opt, captured=cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm._run_epoch(model, Loader((host[i % 3], targets) for i in range(steps)), crit, "cuda", kw, -1, "epoch", opt, captured=cap)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(images_per_s=round(32 * steps / dt, 1), ms_per_step=round(1e3 * dt / steps, 3), steps=steps, batch=32,
                host_bytes_per_step=int(host[0].numel() * 4), captured=bool(cap.captured), prefetch="second stream, one batch ahead")


def measure_nms(dev):
    """Class-aware NMS on BASELINE config 5's tensor (8 x 84 x 33600, fp16) on the device: ms per image."""
    from src.utils.model_utils import non_max_suppression
    pred = config5_nms_tensor(bs=8).half().to(dev)
    out = non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, nc=80)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, nc=80)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    return dict(ms_per_image=round(ms / 8, 3), ms_per_batch=round(ms, 3), images=8,
                candidates_per_image=int((pred[:, 4:].amax(1) > 0.25).sum()) // 8, kept_per_image=sum(o.shape[0] for o in out) // 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (BASELINE config 2/3: 32)")
    ap.add_argument("--res", type=int, default=640)
    ap.add_argument("--preset", choices=sorted(PRESETS), default="s", help="other presets: extra data points, not the metric")
    ap.add_argument("--precision", choices=["bfloat16", "float16", "float32"], default="bfloat16")
    ap.add_argument("--nc", type=int, default=80)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra data points (preset l at 16 images, config-5 NMS)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if os.environ.get("BENCH_ONE_DEVICE"):       # rehearsal of the N > 1 control flow on a one-GPU box (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=dev)
        else:
            dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)

    from src.hipops import ops
    from src.model.losses import PackedTargets, YoloDFLQFLoss
    from src.model.model_builder import Model
    from src.training.graph_step import TrainStepRunner

    nc = args.nc
    torch.manual_seed(0)                      # identical initial weights on every rank (DDP broadcasts rank 0's)
    model = Model(**PRESETS[args.preset], num_classes=nc).to(dev).train()
    if world > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    crit = YoloDFLQFLoss(num_classes=nc, lambda_box=1.5, lambda_cls=1.0)
    # one-launch AdamW (src/training/fused_adamw.py); BENCH_TORCH_ADAMW=1 selects torch's fused
    # multi-tensor AdamW (12 launches per step) for A/B runs
    if os.environ.get("BENCH_TORCH_ADAMW"):
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True, fused=True)
    else:
        from src.training.fused_adamw import HipAdamW
        opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    img, gts = synthetic_batch(args.batch, args.res, nc, 1234 + rank, dev)
    packed = PackedTargets(gts, dev)
    runner = TrainStepRunner(model, crit, opt, args.precision, use_graph=not args.no_graph,
                             grad_comm_dtype=(None if os.environ.get("BENCH_COMM_DTYPE") == "fp32" else torch.bfloat16)
                             if world > 1 else None)
    runner.capture(img, packed)

    for _ in range(args.warmup):
        runner.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    final_loss = float(loss)
    if rank == 0:
        print(f"[bench] {args.steps} steps in {dt:.4f} s -> {args.batch * world * args.steps / dt:.1f} img/s", file=sys.stderr, flush=True)

    # ---- RCCL evidence for the N > 1 line: who took part in the exchange, on which devices, with which buckets
    rccl = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        ident = f"{os.uname().nodename}/{local}/{getattr(props, 'uuid', '') or getattr(props, 'pci_bus_id', '')}"
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                      # on the device, over the bench's own backend: must count every rank
        rccl = dict(backend=dist.get_backend(), ranks=world, ranks_counted_by_all_reduce=int(ones.item()),
                    unique_devices=len(set(idents)),
                    bucket_bytes=[int(f.numel() * f.element_size()) for f in runner.buckets.flats] if runner.buckets and runner.buckets.flats else None,
                    comm_dtype=str(runner.buckets.flats[0].dtype).replace("torch.", "") if runner.buckets and runner.buckets.flats else None,
                    collective="all_reduce AVG per bucket, async beside the next backward stage" if runner.avg_in_collective else "all_reduce SUM per bucket + 1/world in the unpack",
                    graphs=dict(A=runner.graph is not None, B=runner.graph_b is not None, C=runner.graph2 is not None))

    roofline, groups_out, roofline_hbm, roofline_layerwise = None, None, None, None
    if rank == 0 and not args.no_roofline:
        timer = KernelTimer(ops).install()
        from src.hipops import functions as F_
        F_.OVERLAP_WGRAD = False                     # time every leaf alone (in the graph wgrad runs beside dgrad
        two_streams, F_.HEAD_TWO_STREAMS = F_.HEAD_TWO_STREAMS, False      # and the head's class branches beside the box ones)
        del loss                                     # the replayed step's loss node is not kept alive across the eager step
        runner.loss = None
        st = getattr(runner, "stream", None) or torch.cuda.current_stream()
        try:
            torch.cuda.synchronize()
            # on the stream the step was warmed up and captured on (the parameters' AccumulateGrad nodes live there)
            with torch.cuda.stream(st):
                torch.cuda._sleep(int(1.5e9))        # park the GPU so the host queues the whole step ahead:
                #                                      event pairs then bracket kernel time, not launch latency
                opt.zero_grad(set_to_none=True)
                runner._fwd_bwd(img, packed)         # one instrumented eager fwd+loss+bwd (same kernels and shapes as
                #                                      the graph; no collective: only rank 0 runs this)
            groups = timer.summary()
        finally:
            timer.remove()
            F_.OVERLAP_WGRAD = True
            F_.HEAD_TWO_STREAMS = two_streams
        groups_out = {k: dict(ms=round(v["ms"], 3), launches=v["launches"],
                              tflops=round(v["flops"] / v["ms"] / 1e9, 1) if v["flops"] else None,
                              gbs=round(v["bytes"] / v["ms"] / 1e6, 1) if v["bytes"] else None,
                              frac_of_own_roof=round(v["ideal_ms"] / v["ms"], 4) if v["ideal_ms"] else None)
                      for k, v in groups.items()}
        dom = max((k for k in groups if groups[k]["flops"] > 0), key=lambda k: groups[k]["ms"])
        g = groups[dom]
        ach = g["flops"] / g["ms"] / 1e9
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (they cannot be taken live): the committed
        # measurement of this kernel group is attached with its source, or null when the file is absent
        traffic, traffic_src = None, None
        for name in ("r3_final_pmc.json", "r2_final_pmc.json"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", name)))
                if pm.get("kernel") == dom:
                    traffic, traffic_src = pm["hbm_bytes_per_launch"], pm["source"]
                    break
            except (OSError, ValueError, KeyError):
                pass
        roofline = dict(bound="mfma", kernel=dom, achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_src,
                        launches=g["launches"],
                        avg_launch_us=round(1e3 * g["ms"] / g["launches"], 2),
                        algorithmic_gflop_per_launch=round(g["flops"] / g["launches"] / 1e9, 3))

        # the conv group priced layer by layer against the larger of its MFMA time (2.5 PF) and its HBM time
        # (8 TB/s spec; the 6.29 TB/s a stream kernel measures on this part beside it): sum of ideals / sum measured
        conv_groups = [groups[k] for k in groups if groups[k]["flops"] > 0]
        ideal = sum(v["ideal_ms"] for v in conv_groups)
        ideal_m = sum(v["ideal_ms_hbm_measured"] for v in conv_groups)
        meas = sum(v["ms"] for v in conv_groups)
        roofline_layerwise = dict(bound="max(mfma, hbm) per launch", kernels="conv fwd + dgrad + wgrad leaves",
                                  launches=sum(v["launches"] for v in conv_groups), measured_ms=round(meas, 3),
                                  ideal_ms=round(ideal, 3), frac=round(ideal / meas, 4),
                                  ideal_ms_at_measured_hbm=round(ideal_m, 3), frac_at_measured_hbm=round(ideal_m / meas, 4),
                                  fwd_dgrad_frac=round(g["ideal_ms"] / g["ms"], 4),
                                  peaks=dict(mfma_tflops=MFMA_BF16_PEAK_TFLOPS, hbm_gbs=HBM_PEAK_GBS, hbm_measured_gbs=HBM_MEASURED_GBS))

        # the step as a whole is HBM-bound: the same measurement for the largest bandwidth-bound kernel group
        hb = max((k for k in groups if groups[k]["flops"] == 0 and groups[k]["bytes"] > 0), key=lambda k: groups[k]["ms"])
        gh = groups[hb]
        ach_h = gh["bytes"] / gh["ms"] / 1e6
        roofline_hbm = dict(bound="hbm", kernel=hb, achieved=round(ach_h, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach_h / HBM_PEAK_GBS, 4), launches=gh["launches"],
                            avg_launch_us=round(1e3 * gh["ms"] / gh["launches"], 2),
                            algorithmic_mb_per_launch=round(gh["bytes"] / gh["launches"] / 1e6, 2))

    # ---- extra data points observed by the driver (not the metric): BASELINE config 4's model per GPU and config 5's NMS
    extra = None
    hip_graph, opt_in_graph = runner.graph is not None, runner.opt_in_graph
    if rank == 0 and world == 1 and not args.no_extra and args.preset == "s":
        del runner, opt, model, packed
        torch.cuda.empty_cache()
        extra = {}
        try:
            extra["preset_l_640_bf16_16img"] = measure_preset("l", 16, args.res, nc, dev, steps=20, warmup=5)
            extra["nms_config5_fp16_8img"] = measure_nms(dev)
            extra["inference_l_1280_fp16_fused"] = measure_inference(dev)
            extra["inference_one_image_fp16_fused"] = {p: measure_single_image(dev, p) for p in ("s", "l")}
            extra["preset_s_640_bf16_32img_deterministic_mode"] = measure_preset("s", args.batch, args.res, nc, dev, steps=10, warmup=3, deterministic=True)
            extra["preset_l_fsdp2_bf16_16img"] = measure_fsdp2("l", 16, args.res, nc, dev, steps=8, warmup=3)
            extra["preset_l_sharded_captured_bf16_16img"] = measure_sharded("l", 16, args.res, nc, dev, steps=20, warmup=5)
            extra["train_loop_from_host_memory_s_32img"] = measure_epoch_loop(dev)
        except Exception as e:                     # never lose the headline line to an extra
            extra["error"] = repr(e)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.train_step import time_cpu_nms, time_cpu_steps
        # launched like slurm/distributed_training_cpu.sbatch:87-91: one rank, gloo, DDP gradient averaging, all host threads
        # (the reference's thread rule for one rank = torch's default = every host core; on a many-core box that
        # oversubscribes the small layers, so the same steps on 16 threads -- one GPU's CPU share -- are timed beside it)
        def best_of(preset, res, steps, budgets):
            """The same steps at torch's default thread count (the reference's rule for one rank: every host core) and on
            16 threads (one GPU's CPU share of the box); `value` = the faster of the two, the other stays as a sub-key."""
            runs = [time_cpu_steps(preset, res, batch=2, steps=steps, warmup=1, budget_s=budgets[0]),
                    time_cpu_steps(preset, res, batch=2, steps=steps, warmup=1, budget_s=budgets[1], threads=16)]
            for r in runs:
                r["value"] = round(r["value"], 3)
            best = max(runs, key=lambda r: r["value"])
            out = dict(best)
            out["thread_counts_tried"] = {str(r["cores"]): dict(value=r["value"], unit="images/s", sample=r["sample"]) for r in runs}
            return out
        cpu = best_of("s", args.res, 8, (12.0, 10.0))
        cpu["config1_n320_fp32_batch2"] = best_of("n", 320, 12, (4.0, 3.0))        # BASELINE config 1
        cpu["nms_config5"] = time_cpu_nms(images=2, pred=config5_nms_tensor(bs=2))

    if rank == 0:
        gb = args.batch * world
        out = dict(metric="images/sec (640x640 bf16)", value=round(gb * args.steps / dt, 2), unit="images/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype={"bfloat16": "bf16", "float16": "f16", "float32": "f32"}[args.precision], data="synthetic",
                   config=dict(workload=f"preset {args.preset} (width {PRESETS[args.preset]['width'][1]}..{PRESETS[args.preset]['width'][5]}, depth {PRESETS[args.preset]['depth'][0]}) {args.res}x{args.res} train step "
                                        f"(fwd+DFL/QFL loss+bwd+grad sync+AdamW), COCO-80 synthetic, {args.batch} img/GPU",
                               global_batch=gb, parallelism=f"dp{world}", hip_graph=hip_graph,
                               optimizer_in_graph=opt_in_graph, final_loss=round(final_loss, 5), rccl=rccl),
                   roofline=roofline, roofline_layerwise=roofline_layerwise, roofline_hbm=roofline_hbm,
                   roofline_groups=groups_out, cpu_baseline=cpu, extra=extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                 # rank 0 may still be timing its instrumented step: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
