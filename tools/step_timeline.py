"""One replayed step out of a rocprofv3 kernel trace (csv): the kernels between two consecutive loss `k_back` launches, in
start order, with queue, duration, grid and the per-queue busy time -- to see which launches carry the main stream.
    python tools/step_timeline.py kernel_trace.csv [step_index] [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_back" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
seg = rows[marks[k] + 1:marks[k + 1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
span = (int(seg[-1]["End_Timestamp"]) - t0) / 1e3


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:70]


busy = collections.Counter()
byname = collections.defaultdict(lambda: [0.0, 0])
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy[r["Queue_Id"]] += d
    e = byname[(r["Queue_Id"], short(r["Kernel_Name"]))]
    e[0] += d
    e[1] += 1
print(f"step {k}: {len(seg)} kernels, span {span:.1f} us; busy per queue: " + ", ".join(f"q{q}: {b:.0f} us" for q, b in busy.items()))
for (q, n), (d, c) in sorted(byname.items(), key=lambda t: -t[1][0])[:top]:
    print(f"  q{q} {d:8.1f} us {c:4d} x {d / c:7.1f}  {n}")
