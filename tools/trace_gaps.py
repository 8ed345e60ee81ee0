#!/usr/bin/env python
"""From a rocprofv3 --kernel-trace CSV: per-kernel busy time vs idle gaps between consecutive dispatches on the GPU
(are launch gaps worth a hipGraph?).

    python tools/trace_gaps.py <kernel_trace.csv> [min_kernels_per_window]
Prints one JSON object: total kernel time, total gap time (gaps < 1 ms only: larger ones are host pauses between phases),
gap histogram, per-dispatch mean gap.
"""
import csv
import json
import sys


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    busy = sum(e - s for s, e, _ in rows)
    gaps = []
    for (s0, e0, _), (s1, e1, _) in zip(rows, rows[1:]):
        g = s1 - e0
        if 0 <= g < 1_000_000:
            gaps.append(g)
    gaps.sort()
    n = len(gaps)
    out = {"dispatches": len(rows), "kernel_time_ms": round(busy / 1e6, 3), "gap_time_ms": round(sum(gaps) / 1e6, 3),
           "gap_fraction_of_busy": round(sum(gaps) / max(busy, 1), 4),
           "gap_us_mean": round(sum(gaps) / max(n, 1) / 1e3, 2),
           "gap_us_p50": round(gaps[n // 2] / 1e3, 2) if n else None,
           "gap_us_p90": round(gaps[int(n * 0.9)] / 1e3, 2) if n else None,
           "overlapping_pairs": sum(1 for (s0, e0, _), (s1, e1, _) in zip(rows, rows[1:]) if s1 < e0)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
