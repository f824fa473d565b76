"""Forward chain / backward main chain / side stream of one replayed step from a rocprofv3 kernel trace: the step is cut at
the optimizer launch (k_adamw); before it = backward (main chain on one queue, weight gradients on the other), after it =
forward + loss.    python tools/chain_breakdown.py kernel_trace.csv [top]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
marks = [i for i, r in enumerate(rows) if "k_back" in r["Kernel_Name"]]
k = len(marks) - 3
seg = rows[marks[k] + 1:marks[k + 1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
ia = next(i for i, r in enumerate(seg) if "k_adamw" in r["Kernel_Name"] and "tick" not in r["Kernel_Name"])
bwd, fwd = seg[:ia + 1], seg[ia + 1:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:64]


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def report(name, part):
    if not part:
        return
    span = (int(part[-1]["End_Timestamp"]) - int(part[0]["Start_Timestamp"])) / 1e3
    qs = collections.Counter()
    for r in part:
        qs[r["Queue_Id"]] += dur(r)
    print(f"== {name}: {len(part)} kernels, span {span:.0f} us; busy " + ", ".join(f"q{q} {b:.0f}" for q, b in qs.items()))
    by = collections.defaultdict(lambda: [0.0, 0])
    for r in part:
        e = by[(r["Queue_Id"], short(r["Kernel_Name"]))]
        e[0] += dur(r); e[1] += 1
    for (q, n), (d, c) in sorted(by.items(), key=lambda t: -t[1][0])[:top]:
        print(f"  q{q} {d:8.1f} us {c:4d} x {d / c:7.1f}  {n}")


report("backward (+ optimizer)", bwd)
report("forward + loss", fwd)
