"""-m gpu: the driver's contract on bench.py -- one JSON line on stdout and nothing else, the keys and types the round
prompt names, the roofline object consistent with itself.  A short run (3 steps, no CPU baseline, no extras)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {r.stdout[:500]}"
    d = json.loads(lines[0])
    assert d["metric"] == "images/sec (640x640 bf16)" and d["unit"] == "images/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"] and d["config"]["global_batch"] == 32
    assert d["config"]["hip_graph"] is True and d["config"]["rccl"] is None
    # value and ms_per_step describe the same timed region
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    assert 1000 < d["value"] < 20000, d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.02 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0
    lw = d["roofline_layerwise"]
    assert 0.05 < lw["frac"] < 1.0 and lw["ideal_ms"] < lw["measured_ms"]
    groups = d["roofline_groups"]
    for name in ("bn_forward(normalize+act)", "bn_backward(reduce+apply)", "conv_mfma(fwd+dgrad)", "wgrad_mfma", "loss"):
        assert name in groups and groups[name]["launches"] > 0, name
    assert d["roofline_hbm"]["bound"] == "hbm" and d["roofline_hbm"]["achieved"] < d["roofline_hbm"]["peak"]
    assert d["cpu_baseline"] is None and d["extra"] is None          # skipped by the flags of this run
