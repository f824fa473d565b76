import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in [0, 1, 2, 4, 8, 16, 3, 5, 7, 12, 15, 31]:
    env = dict(os.environ, YOLO_CONV_DBG=str(d), REPS="10")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_ablate.py")], env=env, capture_output=True, text=True).stdout
    us = [l.split(":")[1].split("us")[0].strip() for l in out.splitlines() if l.startswith("dbg")]
    print(f"dbg={d:2d}  " + "  ".join(f"{float(u):7.1f}" for u in us), flush=True)
