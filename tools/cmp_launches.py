#!/usr/bin/env python3
"""tools/cmp_launches.py a.csv b.csv : per-layer-shape time of two `bench.py --dump-launches` files."""
import csv, collections, sys
def agg(f):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = (r['kind'], r['H'], r['cin'], r['cout'])
        d[k] = d.get(k, 0.0) + float(r['ms'])
    return d
o, b = agg(sys.argv[1]), agg(sys.argv[2])
print('total %.2f %.2f %+.1f%%' % (sum(o.values()), sum(b.values()), 100 * (sum(b.values()) / sum(o.values()) - 1)))
for k in o:
    print(k, '%.2f %.2f %+.1f%%' % (o[k], b.get(k, 0), 100 * (b.get(k, 0) / o[k] - 1)))
