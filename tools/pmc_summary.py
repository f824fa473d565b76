"""Summarise rocprofv3 --pmc counter_collection csvs: mean counter value per (kernel, grid)."""
import csv, sys, collections, re, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        if sys.argv[1] not in n: continue
        key = (n[20:62], r.get('Grid_Size', r.get('Grid_Size_X', '')))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, cs in agg.items():
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
