"""Replayed inference: forward + decode of `Model.inference` (reference src/model/model_builder.py:115-139) as one hipGraph
per input shape.

A single 640 x 640 image is ~220 launches of 2 - 30 us: issued one by one the host is the bottleneck (preset s: 2.7 ms per
image eager against 0.94 ms replayed, preset l: 4.5 against 1.9 -- tools/infer_bench.py).  The graph holds the static input,
every intermediate and the decoded (N, 4 + nc, M) output in its own pool; a call copies the image in, replays, and hands the
output to the class-aware NMS (which returns fresh tensors, so nothing a caller keeps aliases the pool).

Validity: the unfused model's kernels read parameters and BatchNorm buffers where they live, so in-place updates
(optimizer steps, load_state_dict) are seen by the next replay.  What a replay cannot follow is a change of STORAGE or of the
module tree: `Model._apply` (.to / .half / .cuda), `fuse()` and `load_weights()` drop the graphs, and so does any change of
the parameters' version counters (a fused conv keeps a packed copy of its frozen weight inside the graph)."""
import collections
import warnings

import torch

from src.hipops import ops


class InferenceGraphs:
    MAX_SHAPES = 4                        # distinct (shape, dtype, autocast) keys kept; the least recently used one goes first

    def __init__(self):
        self.entries = collections.OrderedDict()
        self.disabled = None              # reason, once a capture has failed: that model runs eagerly from then on

    @staticmethod
    def _stamp(model):
        v = 0
        for t in model.parameters():
            v += t._version
        for t in model.buffers():
            v += t._version
        return v

    def run(self, model, image):
        """Decoded predictions (N, 4 + nc, M) for `image` (a device tensor), replayed; None = run eagerly."""
        if self.disabled is not None or not image.is_cuda or model.training or torch.is_grad_enabled():
            return None
        amp = torch.is_autocast_enabled()
        key = (tuple(image.shape), image.dtype, image.device.index, amp, torch.get_autocast_dtype("cuda") if amp else None)
        stamp = self._stamp(model)
        ent = self.entries.get(key)
        if ent is not None and ent["stamp"] != stamp:
            ent = None
            self.entries.clear()                              # weights changed: every graph may hold stale packed copies
        if ent is None:
            try:
                ent = self._capture(model, image, amp)
            except Exception as e:                            # noqa: BLE001 -- a model that cannot be captured stays usable
                self.disabled = f"{type(e).__name__}: {e}"
                warnings.warn(f"inference graph capture failed ({self.disabled}); running eagerly", stacklevel=3)
                return None
            ent["stamp"] = self._stamp(model)                 # (a first eval pass may touch buffers' counters)
            self.entries[key] = ent
            while len(self.entries) > self.MAX_SHAPES:
                self.entries.popitem(last=False)
        self.entries.move_to_end(key)
        ent["x"].copy_(image)
        ent["graph"].replay()
        return ent["y"]

    @staticmethod
    def _capture(model, image, amp):
        nc = model.head.nc

        def fwd(x):
            preds, anchors, strides = model.forward(x)
            return ops.head_decode(preds, anchors, strides, nc)

        x = image.clone()
        cur = torch.cuda.current_stream(image.device)
        side = torch.cuda.Stream(image.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                         # warm-up off the capture: packed weights, anchors, allocator
            for _ in range(2):
                fwd(x)
        cur.wait_stream(side)
        torch.cuda.synchronize(image.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            if amp:                                           # no cast cache: a cached cast would live in the graph's pool
                with torch.autocast("cuda", dtype=torch.get_autocast_dtype("cuda"), cache_enabled=False):
                    y = fwd(x)
            else:
                y = fwd(x)
        return dict(x=x, y=y, graph=g)
