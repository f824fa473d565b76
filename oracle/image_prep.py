"""CPU restatement of the reference's image transforms (test infrastructure), src/data/transforms.py:4-24:
ToImage -> RandomHorizontalFlip -> Resize((S, S)) -> ColorJitter -> ToDtype(float32, scale=True) -> Normalize, with the
random decisions passed in.  torchvision (0.24.1, environment.yml:29) is absent from /root/reference and from this image:
PARITY UNPINNED for its arithmetic -- the steps below restate torchvision.transforms.v2.functional as published
(_blend, _rgb_to_grayscale_image, adjust_{brightness,contrast,saturation,hue}_image, to_dtype_image) and the bilinear
antialias resize of torch (aten UpSampleKernel: separable triangle filter, support = scale when shrinking), which
tests/test_oracle_golden.py pins against torch.nn.functional.interpolate(antialias=True) -- the op torchvision calls."""
import numpy as np
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def aa_weights(in_size, out_size):
    """-> list of (lo, weights float32) per output index (torch _upsample_bilinear2d_aa)."""
    scale = np.float32(in_size) / np.float32(out_size)
    support = scale if scale >= 1 else np.float32(1)
    inv = np.float32(1) / scale if scale >= 1 else np.float32(1)
    out = []
    for i in range(out_size):
        center = scale * (np.float32(i) + np.float32(0.5))
        lo = max(0, int(center - support + np.float32(0.5)))
        n = min(in_size, int(center + support + np.float32(0.5))) - lo
        x = np.abs((np.arange(n, dtype=np.float32) + np.float32(lo) - center + np.float32(0.5)) * inv)
        w = np.where(x < 1, np.float32(1) - x, np.float32(0)).astype(np.float32)
        out.append((lo, w / w.sum(dtype=np.float32)))
    return out


def resize_aa(img_hwc_f32, size):
    """float32 (H, W, C) -> (size, size, C): rows then columns of the separable antialiased bilinear filter."""
    h, w, c = img_hwc_f32.shape
    wy, wx = aa_weights(h, size), aa_weights(w, size)
    tmp = np.empty((h, size, c), np.float32)
    for ox, (lo, ww) in enumerate(wx):
        tmp[:, ox] = np.tensordot(img_hwc_f32[:, lo:lo + len(ww)], ww, axes=([1], [0]))
    out = np.empty((size, size, c), np.float32)
    for oy, (lo, ww) in enumerate(wy):
        out[oy] = np.tensordot(tmp[lo:lo + len(ww)], ww, axes=([0], [0]))
    return out


def _gray(x):       # (.., 3) float32 of uint8 levels -> floor(gray)
    return np.floor(np.float32(0.2989) * x[..., 0] + np.float32(0.587) * x[..., 1] + np.float32(0.114) * x[..., 2])


def _blend(a, b, ratio):
    return np.floor(np.clip(a * np.float32(ratio) + b * np.float32(1.0 - ratio), 0, 255)).astype(np.float32)


def _hue(x, f):
    r, g, b = (x[..., i] * np.float32(1 / 255) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    eqc = maxc == minc
    cr = maxc - minc
    s = cr / np.where(eqc, np.float32(1), maxc)
    crd = np.where(eqc, np.float32(1), cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = np.where(maxc == r, bc - gc, 0)
    hg = np.where((maxc == g) & (maxc != r), 2 + rc - bc, 0)
    hb = np.where((maxc != g) & (maxc != r), 4 + gc - rc, 0)
    h = np.fmod((hr + hg + hb) / 6 + 1, 1).astype(np.float32)
    h = h + np.float32(f)
    h = h - np.floor(h)
    v = maxc
    h6 = h * 6
    fi = np.floor(h6)
    ff = (h6 - fi).astype(np.float32)
    i = fi.astype(np.int64) % 6
    p = np.clip(v * (1 - s), 0, 1)
    q = np.clip(v * (1 - s * ff), 0, 1)
    t = np.clip(v * (1 - s * (1 - ff)), 0, 1)
    ro = np.choose(i, [v, q, p, p, t, v])
    go = np.choose(i, [t, v, v, q, p, p])
    bo = np.choose(i, [p, p, t, v, v, q])
    k = np.float32(255.0 + 1.0 - 1e-3)
    return np.stack([np.floor(ro * k), np.floor(go * k), np.floor(bo * k)], -1).astype(np.float32)


def transform_image(img_u8_hwc, size=640, flip=False, order=(), factors=(1.0, 1.0, 1.0, 0.0), mean=MEAN, std=STD):
    """One image through the pipeline.  order: the jitter ops in application order (0 brightness, 1 contrast,
    2 saturation, 3 hue; empty = validation transform); factors indexed by op.  -> float32 (3, size, size)."""
    x = np.asarray(img_u8_hwc).astype(np.float32)
    if flip:
        x = x[:, ::-1]
    x = np.clip(np.floor(resize_aa(np.ascontiguousarray(x), size) + np.float32(0.5)), 0, 255)
    for op in order:
        f = factors[op]
        if op == 0:
            x = _blend(x, np.float32(0), f)
        elif op == 1:
            x = _blend(x, np.float32(_gray(x).mean(dtype=np.float64)), f)
        elif op == 2:
            x = _blend(x, _gray(x)[..., None], f)
        elif op == 3:
            x = _hue(x, f)
    x = x * np.float32(1 / 255)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))


def transform_boxes(boxes_xywh, w, h, size=640, flip=False):
    """torchvision's BoundingBoxes (XYWH, canvas (h, w)) through horizontal flip and resize: (M, 4) float32."""
    b = boxes_xywh.clone().float()
    if flip:
        b[:, 0] = w - (b[:, 0] + b[:, 2])
    b[:, [0, 2]] *= size / w
    b[:, [1, 3]] *= size / h
    return b
