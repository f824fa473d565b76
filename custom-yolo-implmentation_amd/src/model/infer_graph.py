"""Replayed inference: forward + decode of `Model.inference` (reference src/model/model_builder.py:115-139) as one hipGraph
per input shape.

A single 640 x 640 image is ~220 launches of 2 - 30 us: issued one by one the host is the bottleneck (preset s: 2.7 ms per
image eager against 0.94 ms replayed, preset l: 4.5 against 1.9 -- tools/infer_bench.py).  The graph holds the static input,
every intermediate, the decoded (N, 4 + nc, M) tensor and the class-aware NMS on it (its launches have fixed grids: the
candidate capacity, not the candidate count) in its own pool; a call copies the image in, replays, reads the per-image
counts (the one device -> host sync of non_max_suppression) and returns CLONED rows, so nothing a caller keeps aliases the pool.

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
        self.tensors = None
        self.modules = None

    def _stamp(self, model):
        """Sum of the version counters of every parameter and buffer (walking the module tree costs ~0.8 ms on preset s, so
        the tensor list is kept; whatever replaces tensors or modules -- _apply, fuse, load_weights -- drops this object)."""
        if self.tensors is None:
            self.tensors = list(model.parameters()) + list(model.buffers())
        return sum(t._version for t in self.tensors)

    def all_eval(self, model):
        """Whether every module is in eval mode already -- what `model.eval()` would establish, without its walk over the
        module tree on every call (~0.8 ms on preset s; the module list is kept like the tensor list)."""
        if self.modules is None:
            self.modules = list(model.modules())
        return not any(m.training for m in self.modules)

    def run(self, model, image, conf_thres, iou_thres):
        """Detections (list of (n, 6) tensors, as non_max_suppression returns them) for `image` (a device tensor), replayed;
        None = run eagerly."""
        if self.disabled is not None or not image.is_cuda or model.training or torch.is_grad_enabled():
            return None
        amp = torch.is_autocast_enabled()
        if not (0 <= conf_thres <= 1 and 0 <= iou_thres <= 1):
            return None                                       # non_max_suppression raises the reference's assertion
        key = (tuple(image.shape), image.dtype, image.device.index, amp, torch.get_autocast_dtype("cuda") if amp else None,
               float(conf_thres), float(iou_thres))
        stamp = self._stamp(model)
        ent = self.entries.get(key)
        if ent is not None and ent["stamp"] != stamp:
            ent = None
            self.entries.clear()                              # weights changed: every graph may hold stale packed copies
        if ent is None:
            try:
                ent = self._capture(model, image, amp, float(conf_thres), float(iou_thres))
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
        rows, counts, status = ent["y"]
        host = torch.cat((counts, status)).tolist()           # the one device -> host sync, as in non_max_suppression
        if host[-1]:
            raise RuntimeError("yolo_nms: more candidates than the kernel capacity; raise conf_thres")
        rows = rows.clone()
        return [rows[i, :host[i]] for i in range(image.shape[0])]

    @staticmethod
    def _capture(model, image, amp, conf_thres, iou_thres):
        nc = model.head.nc

        def fwd(x):
            preds, anchors, strides = model.forward(x)
            y = ops.head_decode(preds, anchors, strides, nc)
            return ops.nms(y, model.num_classes, conf_thres, iou_thres, None, False, False, 300)

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
