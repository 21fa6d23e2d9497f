#!/usr/bin/env python3
"""Groups the bn_fold launches of a rocprofv3 --kernel-trace CSV by launch shape: python tools/bnfold_trace.py <kernel_trace.csv> [pattern]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else 'bn_fold'
wg = [k for k in rows[0].keys() if 'workgroup' in k.lower() and 'x' in k.lower()] or [k for k in rows[0].keys() if 'workgroup' in k.lower()]
gr = [k for k in rows[0].keys() if 'grid' in k.lower() and 'x' in k.lower()] or [k for k in rows[0].keys() if 'grid' in k.lower()]
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if pat in n:
        d[(n[:24], r[wg[0]], r[gr[0]])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print(k, 'launches', len(v), 'avg us', round(sum(v) / len(v), 1), 'total us', round(sum(v), 1))
