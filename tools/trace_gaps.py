"""Idle time inside one replayed step of a rocprofv3 kernel trace: periods where NO kernel runs, attributed to the
kernel that starts after each gap (cross-stream fork/join latency shows up here)."""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fin = [i for i, r in enumerate(rows) if 'k_back' in r['Kernel_Name']]
seg = rows[fin[-3]:fin[-2]]
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:44], r['Queue_Id']) for r in seg]
span = iv[-1][1] - iv[0][0]
end, idle, gaps = iv[0][0], 0, []
for s, e, n, q in iv:
    if s > end:
        idle += s - end
        gaps.append((s - end, n))
    end = max(end, e)
print(f"kernels {len(iv)}  span {span / 1e3:.1f} us  idle {idle / 1e3:.1f} us in {len(gaps)} gaps (median {statistics.median([g[0] for g in gaps]) / 1e3:.2f} us)")
print("queues:", dict(collections.Counter(q for *_, q in iv)))
c = collections.defaultdict(lambda: [0, 0.0])
for d, n in gaps:
    c[n][0] += 1
    c[n][1] += d / 1e3
for n, (k, t) in sorted(c.items(), key=lambda x: -x[1][1])[:12]:
    print(f"{t:8.1f} us {k:4d}  before {n}")
