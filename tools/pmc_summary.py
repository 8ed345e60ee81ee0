#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel name, mean of each counter per dispatch."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        name = r['Kernel_Name']
        if filt and filt not in name:
            continue
        short = name.split('(')[0][-70:]
        key = (short, r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, ctrs in acc.items():
    n = max(len(v) for v in ctrs.values())
    print(key, 'dispatches=%d' % n)
    for c, v in sorted(ctrs.items()):
        print('    %-32s mean %.4g' % (c, sum(v) / len(v)))
