"""One training step's kernels (between two k_back launches (the last kernel of the loss)) from a rocprofv3 kernel trace."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fin = [i for i, r in enumerate(rows) if 'k_back' in r['Kernel_Name']]
seg = rows[fin[-3]:fin[-2]]
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
print(f'kernels/step {len(seg)}  span {span:.3f} ms  busy {busy:.3f} ms')
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r['Kernel_Name']
    n = n.replace('void (anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
    n = n[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]
    agg[n][0] += 1
    agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f'{t:9.1f} us {c:4d} {t/c:7.1f}  {n}')
