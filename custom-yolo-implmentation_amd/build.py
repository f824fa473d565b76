#!/usr/bin/env python3
"""Build libyolo_hip.so (gfx950) in-tree: hipcc cross-compiles without a GPU.

    python custom-yolo-implmentation_amd/build.py [--force]

Output: custom-yolo-implmentation_amd/src/hipops/libyolo_hip.so (git-ignored, travels with gpurun).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "src", "hipops", "libyolo_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
        "-I", CSRC, "-I", os.path.join(HERE, "..", "include")]
# bit-exact index selection (NMS) and the reference's operation order (loss box decode, cdist form)
NO_CONTRACT = {"loss.hip", "decode_nms.hip", "image_prep.hip"}   # + the uint8-level arithmetic of the input pipeline


def abi_hash(header):
    """63-bit hash of the header's prototypes: compiled into the library (yolo_abi_hash) and checked by the loader, so a
    stale .so next to a newer header is an error at load time instead of shifted arguments at call time."""
    import hashlib
    import re
    protos = re.findall(r"^(?:int|long|size_t)\s+yolo_\w+\s*\([^)]*\)\s*;", open(header).read(), re.M)
    text = "\n".join(p for p in protos if "yolo_abi_hash" not in p)
    return int.from_bytes(hashlib.sha1(text.encode()).digest()[:8], "little") >> 1


def newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "yolo_hip.h"))
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest([src] + hdrs):
            flags = BASE + (["-ffp-contract=off"] if s in NO_CONTRACT else [])
            if s == "abi.hip":
                flags = flags + [f"-DYOLO_ABI_HASH={abi_hash(hdrs[-1])}L"]
            jobs.append((s, [HIPCC] + flags + ["-c", src, "-o", obj]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed on {name}:\n{r.stderr[-4000:]}")
        return name

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for name in ex.map(run, jobs):
            if verbose:
                print(f"[build] compiled {name}", flush=True)
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or not os.path.exists(OUT) or os.path.getmtime(OUT) < newest(objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs,
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print(f"[build] linked {OUT}", flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
