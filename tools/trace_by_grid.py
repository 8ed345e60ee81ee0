#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace (…_kernel_trace.csv) per (kernel, grid size): launches, average / min us.  One kernel
runs many layer shapes in a forward; the grid size tells them apart."""
import csv
import re
import sys
from collections import defaultdict


def main():
    rows = defaultdict(list)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            name = re.sub(r'\(.*', '', name)[:60]
            rows[(name, int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1))].append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    print('%-60s %10s %6s %9s %9s' % ('kernel', 'grid', 'n', 'avg us', 'min us'))
    for (name, grid), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        if pat and not re.search(pat, name):
            continue
        print('%-60s %10d %6d %9.1f %9.1f' % (name, grid, len(v), sum(v) / len(v), min(v)))


if __name__ == '__main__':
    main()
