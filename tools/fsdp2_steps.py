"""Per-step host times of config 4's wrapper route (forward / backward / optimizer + sync) and the time spent in the Python
garbage collector; `after` first runs the extras bench.py runs before it.  Usage: fsdp2_steps.py [after]"""
import sys, time, os
sys.path.insert(0, "custom-yolo-implmentation_amd"); sys.path.insert(0, ".")
import torch, torch.distributed as dist, socket
import bench
from bench import PRESETS, synthetic_batch
from src.model.losses import YoloDFLQFLoss
from src.model.model_builder import Model
from src.training.utils_train import get_optimizer, prepare_fsdp2_model
import gc
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
if len(sys.argv) > 1:
    print(bench.measure_preset("l", 16, 640, 80, dev, steps=5, warmup=2), file=sys.stderr)
    print(bench.measure_inference(dev), file=sys.stderr)
    print(bench.measure_preset("s", 32, 640, 80, dev, steps=5, warmup=2, deterministic=True), file=sys.stderr)
    torch.cuda.empty_cache()
gc_t = [0.0, 0, 0.0]
def _cb(phase, info):
    if phase == "start":
        gc_t[2] = time.perf_counter()
    else:
        gc_t[0] += time.perf_counter() - gc_t[2]; gc_t[1] += 1
gc.callbacks.append(_cb)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
torch.manual_seed(0)
model = Model(**PRESETS["l"], num_classes=80)
model = prepare_fsdp2_model(model=model, device_id=0, config={"precision": "bfloat16"}, world_size=1, device="cuda").train()
opt, _ = get_optimizer(model, lr=1e-4, weight_decay=1e-4, patience=3, factor=0.5)
crit = YoloDFLQFLoss(num_classes=80)
img, gts = synthetic_batch(16, 640, 80, 4321, dev)
ts = []
gc_t[0], gc_t[1] = 0.0, 0
for i in range(40):
    t0 = time.perf_counter()
    opt.zero_grad()
    preds, a, s = model(img)
    loss, ld = crit(preds, gts, a, s)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2))
for i, (f, b, o) in enumerate(ts):
    print(f"step {i:2d}: fwd {f*1e3:6.1f} bwd {b*1e3:6.1f} opt+sync {o*1e3:6.1f} ms", file=sys.stderr)
print(f"garbage collector: {gc_t[1]} collections, {gc_t[0]*1e3:.1f} ms in 40 steps; objects tracked {len(gc.get_objects())}", file=sys.stderr)
if os.environ.get("FSDP2_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(10):
        opt.zero_grad()
        preds, a, s = model(img)
        loss, ld = crit(preds, gts, a, s)
        loss.backward()
        opt.step()
    pr.disable()
    torch.cuda.synchronize()
    import io
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(40)
    print(buf.getvalue(), file=sys.stderr)
st = torch.cuda.memory_stats()
print({k: st[k] for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.peak")}, file=sys.stderr)
dist.destroy_process_group()
