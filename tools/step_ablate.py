import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in [0, 1, 2, 4, 8, 16, 32, 64, 7]:
    env = dict(os.environ, YOLO_CONV_DBG=str(d))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline"],
                         env=env, capture_output=True, text=True).stdout
    try:
        print(f"dbg={d:3d}  ms_per_step {json.loads(out.strip().splitlines()[-1])['ms_per_step']}", flush=True)
    except Exception as e:
        print(d, "failed", out[-300:], flush=True)
