#!/usr/bin/env python
"""Per-layer table from the rocprofv3 --pmc passes of tools/pmc_layers.py.

    python tools/pmc_layers_summary.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]

Every csv holds the same dispatch sequence (same program), so counters of different passes are joined on the ordinal
of the dispatch among the conv kernels of the LAST forward (its launches in layer order: 75, fewer where layers are fused).  The calibration
copy (largest-grid non-conv dispatch after the warm-up) gives bytes per counter unit for a 16-B/lane streaming kernel.
"""
import csv
import json
import sys
from collections import OrderedDict, defaultdict

CONV = ('conv_wino', 'wino44_input_transform_kernel', 'conv_mfma_f32_kernel', 'conv_stem_kernel', 'conv_mfma_bf16_kernel', 'conv_bf16x_kernel',
        'conv_bf16p_kernel', 'conv1x1_bf16r_kernel', 'conv_stem_bf16_kernel', 'conv_stem_mfma_kernel', 'conv_stem_s2_f32_kernel',
        'conv_stem_s2_bf16_kernel', 'conv_resblock64_bf16_kernel')
# the kernels a forward starts with (round 5: the stem may be fused with the conv behind it)
FIRST = ('conv_stem_kernel', 'conv_stem_bf16_kernel', 'conv_stem_mfma_kernel', 'conv_stem_s2_f32_kernel', 'conv_stem_s2_bf16_kernel')


def load(path):
    disp = OrderedDict()        # dispatch id -> (kernel, {counter: value})
    with open(path) as f:
        for r in csv.DictReader(f):
            d = int(r['Dispatch_Id'])
            if d not in disp:
                disp[d] = (r['Kernel_Name'], defaultdict(float))
            disp[d][1][r['Counter_Name']] += float(r['Counter_Value'])
            disp[d][1]['duration_ns'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return [disp[k] for k in sorted(disp)]


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    layers, calib = None, {}
    for p in paths:
        seq = load(p)
        convs = [(n, c) for (n, c) in seq if any(k in n for k in CONV)]
        # the LAST forward: from the last first-layer kernel to the end (75 launches, fewer where layers are fused: the rows
        # are then launches, in layer order)
        starts = [i for i, (n, _) in enumerate(convs) if any(k in n for k in FIRST)]
        last = convs[starts[-1]:] if starts else convs[-75:]
        # round 6: an F(4x4,3x3) layer may be TWO launches (wino44_input_transform_kernel writes V, conv_wino44v_f32_kernel reads
        # it): one row, the counters of both added up (the kernel name keeps both)
        merged = []
        for n, c in last:
            if merged and 'wino44_input_transform_kernel' in merged[-1][0] and 'conv_wino44v' not in merged[-1][0]:
                pn, pc = merged.pop()
                cc = defaultdict(float, pc)
                for k, v in c.items():
                    cc[k] += v
                merged.append((pn.split('(')[0] + ' + ' + n, cc))
            else:
                merged.append((n, c))
        last = merged
        if layers is None:
            layers = [dict(kernel=n.replace('(anonymous namespace)::', '').replace('y3conv::', '').split('(')[0].replace('void ', '')) for n, _ in last]
        for row, (n, c) in zip(layers, last):
            row.update({k: v for k, v in c.items()})
        # calibration: the elementwise copy kernel with the largest counters
        cands = [(n, c) for (n, c) in seq if 'elementwise' in n or 'copyBuffer' in n]
        if cands:
            n, c = max(cands, key=lambda t: sum(t[1].values()))
            calib.update({k: v for k, v in c.items()})
            calib['kernel'] = n[:80]
    res = {"calibration_1GiB_copy": calib, "layers": layers}
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    keys = [k for k in layers[0] if k != 'kernel']
    print('calibration (1 GiB read + 1 GiB written):', {k: v for k, v in calib.items() if k != 'kernel'})
    print('layer kernel ' + ' '.join(keys))
    for i, row in enumerate(layers):
        print(i, row['kernel'][:40], ' '.join('%.4g' % row.get(k, float('nan')) for k in keys))


if __name__ == '__main__':
    main()
