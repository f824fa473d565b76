"""Idle periods of the backward's main queue inside one traced step: gaps between consecutive kernels of the queue that
carries the BatchNorm backward, with what the other queue was running meanwhile.  python tools/queue_gaps.py trace.csv [min_us]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
marks = [i for i, r in enumerate(rows) if "k_back" in r["Kernel_Name"]]
k = len(marks) - 3
seg = rows[marks[k] + 1:marks[k + 1] + 1]
ia = next(i for i, r in enumerate(seg) if "k_adamw" in r["Kernel_Name"] and "tick" not in r["Kernel_Name"])
bwd = seg[:ia + 1]
t0 = int(bwd[0]["Start_Timestamp"])
import collections
mainq = collections.Counter(r["Queue_Id"] for r in bwd if "bwd_apply" in r["Kernel_Name"]).most_common(1)[0][0]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:40]
main = [r for r in bwd if r["Queue_Id"] == mainq]
other = [r for r in bwd if r["Queue_Id"] != mainq]
tot = 0.0; n = 0
for a, b in zip(main, main[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g >= thr:
        tot += g; n += 1
        s, e = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
        run = [short(o["Kernel_Name"]) for o in other if int(o["Start_Timestamp"]) < e and int(o["End_Timestamp"]) > s]
        print(f"{(s - t0) / 1e3:8.1f} us  gap {g:6.1f}  after {short(a['Kernel_Name']):40s} before {short(b['Kernel_Name']):40s} | other: {len(run)} {run[:3]}")
small = sum(max(0.0, (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3) for a, b in zip(main, main[1:]))
print(f"main queue q{mainq}: {len(main)} kernels; gaps >= {thr} us: {n}, {tot:.0f} us; all gaps {small:.0f} us")
