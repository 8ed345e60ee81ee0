#!/usr/bin/env python
"""Where the forward's time above its own bound sits, by kernel family (DESIGN.md section 8's table).

    python tools/gap_table.py [profiles/r04_layers_bs32_416_wino.csv] [--batch 32] [--size 416] [--r03]

Input: the per-layer csv of tools/layer_profile.py (layer, k, stride, cin, cout, ms).  Per layer the bound is
max(algorithmic bytes / 8 TB/s, ISSUED FLOPs / 157.3 TF/s) - bench.py's `whole_forward_frac` accounting: a Winograd
F(2x2,3x3) layer issues 16/36 of the direct-convolution FLOPs, an F(4x4,3x3) layer (round 4: the convs with Cin >= 64 at
this batch, y3_conv_wino44_preferred; --r03: the 128->256 and 512->1024 convs, round 3's rule, for round 3's csv) 36/144
times the padding of its 4x4 tiles.  No GPU needed.
"""
import argparse
import collections
import csv
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def w44_tiles(n, g, mosaic=True):
    """4x4 output tiles of a batch of n g x g maps: image by image, or (round 4's last change: csrc/y3_conv_wino44.hip,
    w44_tiling) the batch as one mosaic of mr x mc images with a zero row / column between neighbours."""
    best = n * (-(-g // 4)) ** 2
    if g % 4 and mosaic:
        for r in range(1, n + 1):
            if n % r == 0:
                best = min(best, (-(-(r * (g + 1) - 1) // 4)) * (-(-(n // r * (g + 1) - 1) // 4)))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv', nargs='?', default=os.path.join(ROOT, 'profiles', 'r06_layers_bs32_416_wino.csv'))
    ap.add_argument('--r03', action='store_true', help="round 3's F(4x4) layer set")
    ap.add_argument('--mosaic', action='store_true', help='the csv was taken with the mosaic tiling of the 13- / 26-grids '
                    '(profiles/r04_layers_bs32_416_wino.csv was not)')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=416)
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = list(csv.DictReader(open(args.csv)))
    table = [(int(r['k']), int(r['stride']), int(r['cin']), int(r['cout']), 1) for r in rows]
    ms = np.array([float(r['ms']) for r in rows])
    flops = bench.conv_flops(table, args.batch, args.size, args.size)
    nbytes = bench.conv_bytes(table, args.batch, args.size, args.size)
    grids = bench.layer_input_grids(table, args.size)
    names, factor = [], []
    for (k, s, cin, cout, _), g in zip(table, grids):
        if k == 3 and s == 1 and cin >= 32:
            if ((cin, cout) in ((128, 256), (512, 1024))) if args.r03 else cin >= 64:      # y3_conv_wino44_preferred at this batch
                names.append('F(4x4,3x3) %d->%d @%d' % (cin, cout, g))
                factor.append(0.25 * w44_tiles(args.batch, g, args.mosaic) * 16 / float(args.batch * g * g))
            else:
                names.append('F(2x2,3x3) %d->%d @%d' % (cin, cout, g))
                factor.append(16.0 / 36.0)
        else:
            names.append('3x3 stride 2' if (k, s) == (3, 2) else ('1x1' if k == 1 else 'stem'))
            factor.append(1.0)
    # round 5: the stem and the stride-2 conv behind it run as ONE kernel (csrc/y3_conv_f32s.hip): the csv then shows ~0 for layer 0
    # and the fused launch under layer 1 - one row of their own, whose bound no longer holds the stem's output (never written / read)
    fused01 = len(ms) > 1 and names[0] == 'stem' and ms[0] < 0.05 * ms[1]
    if fused01:
        names[0] = names[1] = 'stem + 3x3 s2 32->64 (fused)'
        fused = np.zeros(len(ms), int)
        fused[0], fused[1] = 1, 2                    # (what model.layer_fused / y3_net_layer_fused reports on the GPU box)
        nbytes = bench.conv_bytes(table, args.batch, args.size, args.size, 4, fused)
    issued = flops * np.array(factor)
    bound = np.maximum(nbytes / (bench.PEAK_HBM_TBPS * 1e12), issued / (bench.PEAK_FP32_MFMA_TFLOPS * 1e12)) * 1e3
    groups = collections.OrderedDict()
    for i, name in enumerate(names):
        g = groups.setdefault(name, [0, 0.0, 0.0, 0.0])
        g[0] += 1
        g[1] += ms[i]
        g[2] += bound[i]
        g[3] += issued[i]
    print('%-26s %3s %9s %9s %6s %8s %10s' % ('layers', 'n', 'ms', 'bound ms', 'frac', 'gap ms', 'TF/s issued'))
    for name, (n, m, b, iss) in sorted(groups.items(), key=lambda kv: kv[1][2] - kv[1][1]):
        print('%-26s %3d %9.3f %9.3f %6.2f %8.3f %10.1f' % (name, n, m, b, b / m, m - b, iss / m / 1e9))
    print('%-26s %3d %9.3f %9.3f %6.2f %8.3f' % ('whole forward', len(ms), ms.sum(), bound.sum(), bound.sum() / ms.sum(),
                                                 ms.sum() - bound.sum()))


if __name__ == '__main__':
    main()
