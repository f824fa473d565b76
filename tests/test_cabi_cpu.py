"""No-GPU checks of the drop-in boundary: the shared library loads, exports exactly what include/yolo_hip.h
declares, and its host-only queries answer (no kernel is launched here)."""
import ctypes
import os
import re

from src.hipops import lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    protos = lib.parse_header(os.path.join(ROOT, "include", "yolo_hip.h"))
    assert len(protos) >= 40
    so = ctypes.CDLL(lib.SO_PATH)
    missing = [n for n in protos if not hasattr(so, n)]
    assert not missing, missing


def test_every_entry_point_cites_the_reference_interface_it_replaces():
    text = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    blocks = re.split(r"/\* ---- ", text)[1:]
    assert len(blocks) >= 6
    for b in blocks:
        assert re.search(r"\w+\.py:\d+", b.split("*/")[0]), b[:80]


def test_host_side_queries():
    assert lib.query("yolo_conv_kpad", 64, 16, 3, 1, 0, 0) == 160          # 9*16 = 144 -> 160
    assert lib.query("yolo_conv_kpad", 64, 32, 3, 2, 1, 3) == 256          # class 3: 4 taps * 64
    assert lib.query("yolo_conv_dgrad_wbuf_elems", 64, 32, 3, 2) == 32 * (64 + 128 + 128 + 256)
    assert lib.query("yolo_reduce_nblk", 10, 8) == 1 and lib.query("yolo_reduce_nblk", 10 ** 7, 8) <= 2048
    assert lib.query("yolo_nms_capacity", 8400, 80, 0) == 8400 and lib.query("yolo_nms_capacity", 33600, 80, 1) == 131072
    assert lib.query("yolo_loss_workspace_bytes", 2, 2100, 20) >= 2 * 2100 * 16 + 80


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "SO_PATH", str(tmp_path / "nope.so"))
    try:
        lib.load()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the library is absent")
