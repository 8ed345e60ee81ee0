#!/usr/bin/env python
"""Per-layer hipEvent timing of the fused forward plan -> markdown/CSV table (profiles/).

    python tools/layer_profile.py [--batch 32] [--size 416] [--iters 10] [--csv out.csv]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--csv', default=None)
    ap.add_argument('--precision', default='f32', help='f32 | f32_bf16x6 | f32_bf16x3 | bf16')
    a = ap.parse_args()
    import torch
    import yolov3_tensorflow_amd as y3
    import bench
    model = y3.yolov3(80, bench.ANCHORS)
    model.compute_dtype = a.precision
    x = torch.rand((a.batch, a.size, a.size, 3), device='cuda')
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
        bench.random_init(1)
        for _ in range(3):
            model.forward(x)
        ms, table = model.layer_times_ms(x, iters=a.iters)
    flops = bench.conv_flops(table, a.batch, a.size, a.size)
    rows = []
    print('| # | k | s | Cin | Cout | ms | TFLOP/s | %% of fwd |')
    print('|---|---|---|---|---|---|---|---|')
    for i, ((k, s, cin, cout, bn), t, f) in enumerate(zip(table, ms, flops)):
        print('| %d | %d | %d | %d | %d | %.4f | %.1f | %.1f |' % (i, k, s, cin, cout, t, f / t / 1e9, 100 * t / ms.sum()))
        rows.append((i, k, s, cin, cout, t, f / t / 1e9))
    print('total %.3f ms, %.1f TFLOP/s, %.1f img/s' % (ms.sum(), flops.sum() / ms.sum() / 1e9, a.batch / ms.sum() * 1e3))
    for kk in (1, 3):
        sel = np.array([t[0] == kk and t[2] != 3 for t in table])
        print('k=%d: %.3f ms, %.1f TFLOP/s' % (kk, ms[sel].sum(), flops[sel].sum() / ms[sel].sum() / 1e9))
    if a.csv:
        with open(a.csv, 'w') as f:
            f.write('layer,k,stride,cin,cout,ms,tflops\n')
            for r in rows:
                f.write('%d,%d,%d,%d,%d,%.5f,%.2f\n' % r)


if __name__ == '__main__':
    main()
