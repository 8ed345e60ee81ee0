#!/usr/bin/env python
"""Memory-side traffic per launch of the forward's dominant kernel from the byte-weighted gfx950 counters
(TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B: 32-byte units whatever the request size), per layer.

    python tools/pmc_traffic_layers.py <layers.json from tools/pmc_layers_summary.py> <kernel substring> <out.json>

The calibration dispatch of tools/pmc_layers.py (a 1 GiB device copy) must read back as 2^30 bytes read and written;
the factor found is applied (it is 1.000 on this stack).  Unlike FETCH_SIZE (which tallies gfx950's 128-byte requests at
64 bytes) these need no correction.  "DRAM" here = requests the L2 sends to the memory side of the fabric: Infinity
Cache hits are included, so the figures bound HBM traffic from above.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    src, needle, out = sys.argv[1:4]
    d = json.load(open(src))
    cal = d['calibration_1GiB_copy']
    R, W = 'TCC_EA0_RDREQ_DRAM_32B_sum', 'TCC_EA0_WRREQ_WRITE_DRAM_32B_sum'
    fr, fw = (1 << 30) / (cal[R] * 32.0), (1 << 30) / (cal[W] * 32.0)
    rows = []
    for i, l in enumerate(d['layers']):
        rows.append(dict(layer=i, kernel=l['kernel'], read_bytes=int(l[R] * 32 * fr), write_bytes=int(l[W] * 32 * fw)))
    sel = [r for r in rows if any(n in r['kernel'] for n in needle.split(','))]       # (a comma-separated list of substrings)
    assert sel, sorted(set(r['kernel'] for r in rows))
    rd = float(np.mean([r['read_bytes'] for r in sel]))
    wr = float(np.mean([r['write_bytes'] for r in sel]))
    res = {
        "kernel": needle,
        "csrc_sha16": __import__('yolov3_tensorflow_amd.build', fromlist=['csrc_sha16']).csrc_sha16(),
        "launches_counted": len(sel),
        "read_bytes_per_launch": int(rd),
        "write_bytes_per_launch": int(wr),
        "traffic_bytes_per_launch": int(rd + wr),
        "counters": [R, W],
        "calibration": {"copy_bytes": 1 << 30, "read_counter_x32": int(cal[R] * 32), "write_counter_x32": int(cal[W] * 32),
                        "read_factor": round(fr, 5), "write_factor": round(fw, 5)},
        "note": "fabric-side (L2 miss) traffic of one forward at the bench configuration (tools/pmc_layers.py), mean over the "
                "launches of the kernel; Infinity-Cache hits are included, so this bounds HBM bytes from above.",
        "per_layer": rows,
    }
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("kernel", "launches_counted", "read_bytes_per_launch", "write_bytes_per_launch",
                                          "traffic_bytes_per_launch")}))


if __name__ == '__main__':
    main()
