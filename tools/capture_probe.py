"""One-shot probes of the graph-capture failures recorded in DESIGN.md, each in its OWN subprocess under
`python -X faulthandler` so that a death leaves a Python traceback in the log (profiles/r2_capture_probe.log):

  ddp        capture forward + backward of a DistributedDataParallel-wrapped Model the plain way
  ddp_recipe the same following PyTorch's documented recipe for DDP under CUDA graphs (async error handling off, DDP
             constructed on a side stream, 11 eager warm-up iterations on that stream before capture)
  twostream  class branches of the head recorded by AUTOGRAD on a second stream (backward nodes then run there too)
  gc         a pinned-memory owner dropped with the collector ENABLED during capture (the abort of commit 3e4e500)

Usage: capture_probe.py            run all probes (driver), one line of verdict each
       capture_probe.py <name>     run one probe in this process (what the driver spawns)"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-yolo-implmentation_amd"))
NANO = dict(csp=[False, True], depth=[1] * 6, width=[3, 16, 32, 64, 128, 256])


def _batch():
    import torch
    from src.model.losses import PackedTargets
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 160, 160, generator=g).cuda()
    gts = [torch.cat([torch.rand(3, 2, generator=g) * 160, torch.rand(3, 2, generator=g) * 60 + 8,
                      torch.randint(0, 80, (3, 1), generator=g).float()], 1).cuda() for _ in range(2)]
    return img, PackedTargets(gts, img.device)


def _pg():
    import torch.distributed as dist
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)


def probe_ddp(recipe):
    import torch
    from torch.nn.parallel import DistributedDataParallel as DDP
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    torch.cuda.set_device(0)
    _pg()
    img, packed = _batch()
    crit = YoloDFLQFLoss(num_classes=80)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model = Model(**NANO, num_classes=80).cuda().train()
        ddp = DDP(model, device_ids=[0]) if recipe else None
    if not recipe:
        ddp = DDP(model, device_ids=[0])

    def step():
        for p in model.parameters():
            p.grad = None
        preds, a, s = ddp(img)
        loss, _ = crit(preds, packed, a, s)
        loss.backward()
        return loss

    with torch.cuda.stream(side):
        for _ in range(11 if recipe else 2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("[probe] warm-up done, capturing", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        loss = step()
    print("[probe] captured, replaying", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"[probe] OK loss {float(loss):.5f}", flush=True)


def probe_twostream():
    import torch
    from src.hipops import functions as F_
    from src.model.losses import YoloDFLQFLoss
    from src.model.model_builder import Model
    torch.cuda.set_device(0)
    F_.HEAD_TWO_STREAMS = False
    img, packed = _batch()
    model = Model(**NANO, num_classes=80).cuda().train()
    crit = YoloDFLQFLoss(num_classes=80)
    head = model.head
    side2 = torch.cuda.Stream()

    def head_forward(x):            # class branches under an AUTOGRAD-visible stream switch
        cur = torch.cuda.current_stream()
        side2.wait_stream(cur)
        with torch.cuda.stream(side2):
            cls_outs = [head._branch(head.cls[i], x[i]) for i in range(head.nl)]
        box_outs = [head._branch(head.box[i], x[i]) for i in range(head.nl)]
        cur.wait_stream(side2)
        outs = [o for pair in zip(box_outs, cls_outs) for o in pair]
        preds = F_.HeadPack.apply(*outs)
        from src.utils.model_utils import make_anchors_cached
        shapes = tuple((int(o.shape[2]), int(o.shape[3])) for o in outs[::2])
        a, s = make_anchors_cached(shapes, tuple(float(v) for v in head.stride), preds.dtype, preds.device)
        return preds, a, s
    head.forward = head_forward

    def step():
        for p in model.parameters():
            p.grad = None
        preds, a, s = model(img)
        loss, _ = crit(preds, packed, a, s)
        loss.backward()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("[probe] eager two-stream steps done, capturing", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        loss = step()
    print("[probe] captured, replaying", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"[probe] OK loss {float(loss):.5f}", flush=True)


def probe_gc():
    """Without TrainStepRunner.capture's gc.collect() + gc.disable(): drop a pinned-memory owner, force the collector
    to run in the middle of the capture."""
    import gc
    import torch
    torch.cuda.set_device(0)
    x = torch.randn(1 << 20, device="cuda")

    class Owner:                      # pinned host memory held only by a reference cycle -> freed by the collector
        def __init__(self):
            self.buf = torch.zeros(1 << 16).pin_memory()
            self.me = self
            self.dev = self.buf.to("cuda", non_blocking=True)
    Owner()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2
        gc.collect()                  # the pinned block goes back to the caching host allocator here
        y = y + 1
    g.replay()
    torch.cuda.synchronize()
    print("[probe] OK", flush=True)


PROBES = {"ddp": lambda: probe_ddp(False), "ddp_recipe": lambda: probe_ddp(True), "twostream": probe_twostream, "gc": probe_gc}


def main():
    if len(sys.argv) > 1:
        faulthandler.enable(all_threads=True)
        PROBES[sys.argv[1]]()
        return
    for name in PROBES:
        env = dict(os.environ, PYTHONFAULTHANDLER="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if name == "ddp_recipe":
            env["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "0"
        print(f"===== probe {name}", flush=True)
        try:
            r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), name], env=env,
                               capture_output=True, text=True, timeout=240)
            tail = (r.stdout[-1500:] + "\n--- stderr ---\n" + r.stderr[-6000:])
            print(tail)
            print(f"===== probe {name}: exit code {r.returncode}", flush=True)
        except subprocess.TimeoutExpired as e:
            print(f"===== probe {name}: TIMEOUT\n{(e.stderr or b'')[-3000:]}", flush=True)


if __name__ == "__main__":
    main()
