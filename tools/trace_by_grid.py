"""Per (kernel, grid) min/mean duration from a rocprofv3 kernel trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
order = []
for r in rows:
    n = r["Kernel_Name"]
    if "GLOBAL__N_1" not in n and "anonymous" not in n: continue
    k = (n.split("GLOBAL__N_1")[-1][2:34], r["Grid_Size_X"], r["Grid_Size_Y"])
    if k not in agg: order.append(k)
    agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in order:
    v = agg[k]
    print(f"{k[0]:34s} grid {k[1]:>8s}x{k[2]:<3s} n={len(v):3d} min {min(v):8.1f} mean {sum(v)/len(v):8.1f}")
