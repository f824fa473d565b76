"""Summarise a rocprofv3 kernel trace csv: per kernel name count / mean / min duration (us)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)[:int(sys.argv[2]) if len(sys.argv) > 2 else 80]
    agg[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in sorted(agg.items(), key=lambda x: -sum(x[1])):
    print(f"{sum(v):10.1f} us {len(v):5d} mean {sum(v)/len(v):8.1f} min {min(v):8.1f}  {n}")
