"""Training / validation loop and validation decode with the reference's signatures
(src/training/train_model.py:14-142 decode_predictions, :145-384 train).

Same step body as the reference -- zero_grad, autocast only in ddp mode, forward, loss, (scaled) backward,
optimizer step -- on the HIP-backed model; host-side changes only: the three loss scalars arrive with one
device->host copy per step (LazyLossDict) and the six per-epoch scalar all-reduces are two."""
import torch
from torch.amp import GradScaler
from tqdm import tqdm

try:
    from torch.distributed.fsdp.sharded_grad_scaler import ShardedGradScaler
except ImportError:  # pragma: no cover
    ShardedGradScaler = None

from src.hipops import ops
from src.training.distributed_setup import reduce_values
from src.training.metrics import DetectionMetrics
from src.training.utils_train import _is_sharded, checkpoint_states, save_checkpoint


def decode_predictions_packed(preds, anchors, strides, conf_threshold=0.25, top_k=100):
    """Raw head output -> (rows fp32 [N][top_k][6] = cx, cy, w, h, class, score, zero padded; counts int32 [N]), all
    on the device: DFL expectation -> xywh * stride (yolo_head_decode), sigmoid best class, `>= conf`, top-k by score
    (yolo_val_select); no NMS, no per-image launches, no host sync.  Reference :14-142."""
    nc = preds.shape[1] - 64
    y = ops.head_decode(preds, anchors, strides, nc)            # (N, 4+nc, M): boxes + raw class logits
    return ops.val_select(y, nc, conf_threshold, top_k)


def decode_predictions(preds, anchors, strides, conf_threshold=0.25, top_k=100, num_classes=171):
    """Reference signature (:14): list of per-image (k, 5) [cx, cy, w, h, class] fp32 tensors (one host sync for the
    N counts).  Order: anchor order when at most top_k anchors pass, else descending score."""
    rows, count = decode_predictions_packed(preds, anchors, strides, conf_threshold, top_k)
    return [rows[b, :c, :5] if c else torch.zeros(0, 5, device=preds.device) for b, c in enumerate(count.tolist())]


def _make_scaler(precision, distributed_mode, device, rank):
    if precision != "float16":
        if precision == "bfloat16" and rank == 0:
            print("[INFO] Using bfloat16 precision (no scaler needed)")
        return None
    on_gpu = device != "cpu"
    if distributed_mode.startswith("fsdp") and on_gpu and ShardedGradScaler is not None:
        if rank == 0:
            print("[INFO] Initialized ShardedGradScaler for FSDP float16 training")
        return ShardedGradScaler()
    if rank == 0:
        print(f"[INFO] Initialized GradScaler for {distributed_mode} float16 training on {device}")
    return GradScaler("cuda" if on_gpu else "cpu")


class CapturedTraining:
    """Training batches go through `TrainStepRunner`: the step captured once as hipGraphs on static image / target buffers
    that every batch refills (10.4 instead of 13.4 ms per step on preset s: issued launch by launch the step is bound by the
    host -- tools/host_profile.py; 21.7 ms before round 3's cuts of the per-launch Python work).  The default in ddp mode on ONE GPU (`train(captured_step=None)`); with more than one rank it is opt-in
    (`captured_step=True` / config key `training.captured_step: true`) until a multi-GPU run has verified it on hardware --
    the default there is the reference's eager loop under torch's DistributedDataParallel reducer.  `captured_step=False`
    keeps the eager loop everywhere.

    A DistributedDataParallel-wrapped model is stepped through its `.module`: the runner averages the gradients itself
    (two flat buckets, all-reduced beside the backbone's backward: src/training/graph_step.py) in the gradients' own
    dtype (fp32, like the wrapper's reducer: same mean over ranks); `grad_compress="bf16"` (config key
    `training.ddp.grad_compress`) exchanges bf16 buckets instead, DDP's `bf16_compress_hook` -- a deviation of about
    2^-8 relative per element from the reference, never the default.  The wrapper's hooks stay idle because its own
    forward is not called.  (Capturing
    the WRAPPER's forward also works once its logger has stopped sampling, i.e. after 11 eager iterations --
    tools/capture_probe.py, profiles/r2_capture_probe.log -- but then RCCL kernels sit inside the graph; the eager
    collectives between graphs are the conservative choice.)  DDP re-broadcasts the BatchNorm buffers from rank 0 before
    every forward; in training mode nothing reads them, so `sync_buffers()` at the end of the training epoch gives every
    rank the same running statistics DDP would have left (rank 0's).

    The first fitting batch is stepped eagerly (the capture's warm-up).  Whether a batch fits the captured buffers
    (batch size, resolution, box capacity) is decided COLLECTIVELY (all-reduce MIN of a host flag over a gloo side
    group): all ranks replay or all ranks step eagerly, with the same buckets and reduction either way.  Needs a
    capturable optimizer.  float16: no torch GradScaler on this route -- the loss scale, the overflow check, the skipped or
    unscaled update and the scale's growth / backoff (GradScaler's rules and defaults, reference :195-208,247-253) run on the
    device inside the captured step (`DeviceGradScaler`, needs `HipAdamW`)."""

    def __init__(self, model, criterion, optimizer, precision, grad_compress=None):
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel as DDP
        self.wrapper = model
        self.inner = model.module if isinstance(model, DDP) else model
        self.criterion, self.optimizer, self.precision = criterion, optimizer, precision
        if grad_compress not in (None, "", "none", "bf16", "bfloat16"):
            raise ValueError(f"grad_compress: expected 'bf16' or nothing, got {grad_compress!r}")
        self.comm_dtype = torch.bfloat16 if grad_compress in ("bf16", "bfloat16") else None
        self.runner, self.dirty, self.captured = None, False, False
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.flag_group = None
        # (a wrapper built elsewhere on the default stream cannot be captured -- see prepare_ddp_model; TrainStepRunner.capture
        # measures that on its warm-up step and keeps stepping eagerly then: no flag to trust)
        from src.training.fused_adamw import HipAdamW
        self.usable = precision in ("bfloat16", "float32", "float16") and \
            all(g.get("capturable", False) for g in optimizer.param_groups) and \
            all(type(p) is torch.nn.Parameter and p.is_cuda for p in self.inner.parameters()) and \
            (precision != "float16" or (isinstance(optimizer, HipAdamW) and len(optimizer.param_groups) == 1))

    def _all_fit(self, fits):
        """AND of the ranks' `fits` flags (host side: no device sync)."""
        if self.world == 1:
            return fits
        import torch.distributed as dist
        if self.flag_group is None:         # first call, same point of the loop on every rank: a CPU side group
            self.flag_group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
        t = torch.tensor([1 if fits else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.flag_group)
        return bool(t.item())

    def step(self, images, boxes):
        """-> LazyLossDict of the step, or None (run this batch in the caller's eager loop: unusable configuration)."""
        from src.model.losses import LazyLossDict, StaticTargets
        from src.training.graph_step import TrainStepRunner
        if not self.usable:
            return None
        if self.runner is None:
            self.runner = self._make_runner()
        if not self.captured:
            cap = 128 * len(boxes)
            fits = sum(int(b.shape[0]) if b.numel() else 0 for b in boxes) <= cap
            if self._all_fit(fits):
                self.runner.capture_for_batches(images, boxes, warmup=1)    # one eager step on this batch, then the capture
                self.captured = True
                return LazyLossDict(self.runner.warm_scalars)
            _, ld = self.runner._eager_step(images, [b.to(images.device) for b in boxes])
            return ld
        if self._all_fit(self.runner.fits(images, boxes)):
            if self.dirty:      # an eager fallback step made optimizer / buckets rebuild the tables the captured launches read
                for obj in (self.optimizer, self.runner.buckets, getattr(self.runner, "pack", None)):
                    restore = getattr(obj, "restore_capture", None)
                    if restore is not None:
                        restore()
                self.dirty = False
            self.runner.step_batch(images, boxes)
            return LazyLossDict(self.runner.scalars)
        # some rank's batch does not fit the captured buffers: every rank takes the runner's eager step
        self.dirty = True
        _, ld = self.runner._eager_step(images, [b.to(images.device) for b in boxes])
        return ld

    def _make_runner(self):
        from src.training.graph_step import TrainStepRunner
        return TrainStepRunner(self.inner, self.criterion, self.optimizer, self.precision, use_graph=True,
                               grad_comm_dtype=self.comm_dtype)

    def sync_buffers(self):
        """DistributedDataParallel(broadcast_buffers=True) semantics for the BatchNorm buffers: rank 0's values on every
        rank (one flat broadcast per dtype) -- call before evaluating or checkpointing."""
        if self.world == 1:
            return
        import torch.distributed as dist
        by_dtype = {}
        for b in self.inner.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, 0)
            off = 0
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()


class ShardedTraining(CapturedTraining):
    """`--mode fsdp2` with `training.fsdp2.native_shard: true`: training batches go through `ShardedStepRunner`
    (src/training/sharded_step.py) on the static buffers every batch refills.  Same collective fit decision and eager
    fallback as CapturedTraining; every rank keeps its own BatchNorm statistics, as under FSDP (no buffer broadcast)."""

    def __init__(self, model, criterion, optimizer, precision):
        super().__init__(model, criterion, optimizer, precision)
        native = model._native_shard
        self.shard = native["state"]
        self.usable = self.shard is not None and optimizer is native.get("optimizer") and precision in ("bfloat16", "float32")

    def _make_runner(self):
        from src.training.sharded_step import ShardedStepRunner
        return ShardedStepRunner(self.inner, self.criterion, precision=self.precision, use_graph=True, shard=self.shard,
                                 optimizer=self.optimizer)

    def sync_buffers(self):
        return


def _run_epoch(model, loader, criterion, device, autocast_kw, rank, desc, optimizer=None, scaler=None, metrics=None,
               conf_threshold=0.25, num_classes=171, on_step=None, captured=None):
    sums = [0.0, 0.0, 0.0]
    dev_sums = [None]
    if device != "cpu" and torch.cuda.is_available():
        from src.data.data_loader import DevicePrefetcher
        loader = DevicePrefetcher(loader, device)            # the next batch's upload / transform beside this batch's step
    bar = tqdm(loader, desc=desc, disable=(rank != 0))

    def account(i, loss_dict):
        scalars = getattr(loss_dict, "_scalars", None)
        if scalars is not None and scalars.is_cuda:
            # The step's three scalars stay on the device: reading them here would make the host wait for every step (the
            # reference's three .item() calls do), and the next batch's upload could no longer run beside this step.  They
            # are summed on the device in double; the progress bar catches up every tenth step.
            dev_sums[0] = scalars.double() if dev_sums[0] is None else dev_sums[0] + scalars.double()
            if rank == 0 and i % 10 == 9:
                cur = dev_sums[0].tolist()
                bar.set_postfix({"Loss": f"{(sums[0] + cur[0]) / (i + 1):.4f}", "Box": f"{(sums[1] + cur[1]) / (i + 1):.4f}",
                                 "Cls": f"{(sums[2] + cur[2]) / (i + 1):.4f}"})
        else:
            for k, key in enumerate(("total_loss", "box_loss", "cls_loss")):
                sums[k] += loss_dict[key]
            bar.set_postfix({"Loss": f"{sums[0] / (i + 1):.4f}", "Box": f"{sums[1] / (i + 1):.4f}",
                             "Cls": f"{sums[2] / (i + 1):.4f}"})
        if on_step is not None:
            on_step(i, loss_dict)

    for i, (images, targets) in enumerate(bar):
        images = images.to(device)
        loss_dict = captured.step(images, [t["boxes"] for t in targets]) if captured is not None else None
        if loss_dict is not None:
            account(i, loss_dict)
            continue
        gt_box = [t["boxes"].to(device) for t in targets]
        if optimizer is not None:
            optimizer.zero_grad()
        with torch.autocast(**autocast_kw):
            preds, anchors, strides = model(images)
            loss, loss_dict = criterion(preds, gt_box, anchors, strides)
        if optimizer is not None:
            if scaler is not None:
                scaler.scale(loss).backward()
                scaler.step(optimizer)
                scaler.update()
            else:
                loss.backward()
                optimizer.step()
        elif metrics is not None:
            # one select + one matching launch per batch; images without ground truth are skipped (reference :326-328)
            rows, count = decode_predictions_packed(preds, anchors, strides, conf_threshold=conf_threshold)
            metrics.update_batch(rows, count, gt_box, skip_empty_targets=True)
        account(i, loss_dict)
    if dev_sums[0] is not None:
        sums = [a + b for a, b in zip(sums, dev_sums[0].tolist())]
    n = max(len(loader), 1)
    return [s / n for s in sums]


def train(model, train_loader, val_loader, optimizer, scheduler, criterion, initial_epoch, num_epochs, device,
          num_classes=171, rank=0, use_wandb=False, wandb_instance=None, log_interval=10,
          checkpoint_dir="experiments/checkpoints", iou_threshold=0.5, conf_threshold=0.25, distributed_mode="ddp",
          precision="float32", captured_step=None, grad_compress=None):
    """`captured_step`: None (default) = the captured step where it applies AND has been verified on hardware (ddp mode on
    one GPU, plain parameters, capturable optimizer), True = also with more than one rank, False = the reference's eager
    loop.  `grad_compress`: None = fp32 gradient exchange (the reference's), "bf16" = bf16 buckets in the captured step."""
    use_amp = precision in ("float16", "bfloat16")
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    want_captured = captured_step is True or (captured_step is None and world == 1)
    captured = CapturedTraining(model, criterion, optimizer, precision, grad_compress=grad_compress) \
        if (want_captured and device != "cpu" and distributed_mode == "ddp") else None
    if getattr(model, "_native_shard", None) is not None:       # fsdp2 with native_shard: the sharded step IS the training path
        captured = ShardedTraining(model, criterion, optimizer, precision)
        if not captured.usable:
            raise RuntimeError("fsdp2.native_shard: build the optimizer with get_optimizer(model, ...) after prepare_fsdp2_model")
    if captured is not None and not captured.usable:
        captured = None
    # the captured route scales fp16 losses on the device; the eager loop uses torch's (Sharded)GradScaler like the reference
    scaler = _make_scaler(precision, distributed_mode, device, rank) if (use_amp and captured is None) else None
    if captured is not None and precision == "float16" and rank == 0:
        print("[INFO] float16: dynamic loss scaling runs on the device inside the captured step (GradScaler's rules)")
    autocast_kw = dict(device_type="cpu" if device == "cpu" else "cuda",
                       dtype=torch.bfloat16 if precision == "bfloat16" else torch.float16,
                       enabled=(distributed_mode == "ddp" and use_amp))       # FSDP modes rely on their MP policy
    metrics = DetectionMetrics(num_classes=num_classes, iou_threshold=iou_threshold)

    for epoch in range(initial_epoch, num_epochs):
        if hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        model.train()

        def log_step(i, ld, epoch=epoch):
            if use_wandb and rank == 0 and wandb_instance is not None and i % log_interval == 0:
                wandb_instance.log({"train/total_loss": ld["total_loss"], "train/box_loss": ld["box_loss"],
                                    "train/cls_loss": ld["cls_loss"], "step": epoch * len(train_loader) + i})

        tr = _run_epoch(model, train_loader, criterion, device, autocast_kw, rank,
                        f"[Epoch {epoch + 1}/{num_epochs}] Training", optimizer, scaler, on_step=log_step,
                        captured=captured)
        if rank != -1:
            tr = reduce_values(tr, average=True)
        if captured is not None:
            captured.sync_buffers()         # what DDP's per-forward buffer broadcast leaves: rank 0's running statistics

        model.eval()
        metrics.reset()
        with torch.no_grad():
            va = _run_epoch(model, val_loader, criterion, device, autocast_kw, rank,
                            f"[Epoch {epoch + 1}/{num_epochs}] Validation", metrics=metrics,
                            conf_threshold=conf_threshold, num_classes=num_classes)
        va = reduce_values(va, average=True)
        md = metrics.compute()
        scheduler.step(va[0])

        # a sharded model's full state is gathered by a collective: every rank takes part, rank 0 writes
        states = checkpoint_states(model, optimizer) if _is_sharded(model) else None
        if rank == 0:
            if use_wandb and wandb_instance:
                wandb_instance.log({"epoch": epoch + 1, "train/epoch_loss": tr[0], "train/epoch_box_loss": tr[1],
                                    "train/epoch_cls_loss": tr[2], "val/epoch_loss": va[0], "val/epoch_box_loss": va[1],
                                    "val/epoch_cls_loss": va[2], "val/precision": md["precision"],
                                    "val/recall": md["recall"], "val/f1_score": md["f1_score"], "val/mAP": md["mAP"],
                                    "lr": optimizer.param_groups[0]["lr"]})
            save_checkpoint(model, optimizer, epoch + 1, va[0], checkpoint_dir=checkpoint_dir, states=states)
            w = tqdm.write
            w("=" * 80)
            w(f"Epoch {epoch + 1}/{num_epochs} Summary:")
            w(f"  Train - Total: {tr[0]:.4f} | Box: {tr[1]:.4f} | Cls: {tr[2]:.4f}")
            w(f"  Val   - Total: {va[0]:.4f} | Box: {va[1]:.4f} | Cls: {va[2]:.4f}")
            w(f"  Metrics - Precision: {md['precision']:.4f} | Recall: {md['recall']:.4f} | F1: {md['f1_score']:.4f} | mAP: {md['mAP']:.4f}")
            w(f"  Detection - TP: {md['true_positives']} | FP: {md['false_positives']} | FN: {md['false_negatives']}")
            w(f"  LR: {optimizer.param_groups[0]['lr']:.6f}")
            w("=" * 80 + "\n")
