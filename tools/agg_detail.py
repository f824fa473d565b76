import re, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for l in open(sys.argv[1]):
    m = re.match(r'\[detail\]\s+([\d.]+) us (\w+)\s+([\d.]+) TF/s\s+(\d+) GB/s (\[.*\]) (\[.*\])', l)
    if not m: continue
    us, name, tf, gb, shapes, ints = m.groups(); us = float(us); ints = eval(ints)
    if name == 'conv_fwd': key = (name, 'k%d s%d' % (ints[1], ints[2]))
    elif name == 'conv_dgrad': key = (name, 'k%d s%d' % (ints[3], ints[4]))
    elif name == 'conv_wgrad': key = (name, 'k%d s%d' % (ints[0], ints[1]))
    else: key = (name, '')
    agg[key][0] += 1; agg[key][1] += us
tot = sum(v[1] for v in agg.values())
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]): print(f'{t:9.1f} us {c:4d}  {k[0]} {k[1]}')
print('total', tot)
