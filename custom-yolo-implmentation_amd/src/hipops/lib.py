"""Loader for libyolo_hip.so: prototypes are parsed from include/yolo_hip.h so the header is the single
source of truth for the C ABI.  There is no fallback: if the library cannot be loaded, every op raises."""
import ctypes
import os
import re

import torch  # noqa: F401  -- must come first: libyolo_hip.so has to bind to the HIP runtime PyTorch loaded,
#                       or its launches would go through a second runtime that does not know torch's streams

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("YOLO_HIP_SO_TMP") or os.path.join(_HERE, "libyolo_hip.so")
HEADER_PATHS = [os.path.join(_HERE, "..", "..", "..", "include", "yolo_hip.h"),
                os.path.join(_HERE, "yolo_hip.h")]

F32, BF16, F16 = 0, 1, 2
ACT_IDENTITY, ACT_SILU = 0, 1

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
       "double": ctypes.c_double, "hipStream_t": ctypes.c_void_p}
_PROTO = re.compile(r"^(int|long|size_t)\s+(yolo_\w+)\s*\(([^)]*)\)\s*;", re.M)


def parse_header(path=None):
    """-> {name: (restype, [argtype, ...], [argname, ...])}"""
    if path is None:
        path = next((p for p in HEADER_PATHS if os.path.exists(p)), None)
        if path is None:
            raise RuntimeError("include/yolo_hip.h not found next to the package")
    protos = {}
    for ret, name, args in _PROTO.findall(open(path).read()):
        types, names = [], []
        for a in [s.strip() for s in args.split(",") if s.strip() and s.strip() != "void"]:
            toks = a.replace("*", " * ").split()
            names.append(toks[-1])
            types.append(ctypes.c_void_p if "*" in toks else _CT[[t for t in toks[:-1] if t != "const"][0]])
        protos[name] = (_CT[ret], types, names)
    return protos


_lib = None
_protos = None


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"libyolo_hip.so is missing ({SO_PATH}); build it with "
                           "`python custom-yolo-implmentation_amd/build.py` -- there is no CPU fallback")
    lib = ctypes.CDLL(SO_PATH)
    _protos = parse_header()
    for name, (ret, types, _) in _protos.items():
        fn = getattr(lib, name)        # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = ret, types
    want = _header_hash()
    if lib.yolo_abi_hash() != want:
        raise RuntimeError(f"libyolo_hip.so was built against another include/yolo_hip.h (ABI hash {lib.yolo_abi_hash()} != "
                           f"{want}): rebuild it with `python custom-yolo-implmentation_amd/build.py` and check that the build succeeds")
    _lib = lib
    return lib


def _header_hash():
    """Same hash as build.py's abi_hash()."""
    import hashlib
    path = next(p for p in HEADER_PATHS if os.path.exists(p))
    protos = re.findall(r"^(?:int|long|size_t)\s+yolo_\w+\s*\([^)]*\)\s*;", open(path).read(), re.M)
    text = "\n".join(p for p in protos if "yolo_abi_hash" not in p)
    return int.from_bytes(hashlib.sha1(text.encode()).digest()[:8], "little") >> 1


def status(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}")


_fn = {}


def _entry(name):
    f = _fn.get(name)
    if f is None:
        f = _fn[name] = getattr(load(), name)
    return f


def call(name, *args):
    """Checked call of an int-returning entry point."""
    rc = _entry(name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}")


def query(name, *args):
    """Unchecked call (size / capacity queries)."""
    return _entry(name)(*args)
