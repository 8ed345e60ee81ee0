#!/usr/bin/env python
"""Per-layer table of a tools/r05_c5.sh run: every 3x3 / 1x1 tile shape's time per layer, the best of them, the product's
choice (-> profiles/r05_bf16_tiles.txt; the cost models of csrc/y3_conv_bf16x.hip / y3_conv_bf16r.hip are fitted to it)."""
import csv
import os
import sys


def load(p):
    return [r for r in csv.DictReader(open(p))] if os.path.exists(p) else None


def main():
    d = sys.argv[1]
    for tag, pre_x, pre_r in (('bs=16', 'x_', 'r_'), ('bs=8', 'x8_', 'r8_')):
        base = load(os.path.join(d, 'default.csv' if tag == 'bs=16' else 'default_bs8.csv'))
        if base is None:
            continue
        xs = {t: load(os.path.join(d, pre_x + t + '.csv')) for t in 'ABCDE'}
        rs = {t: load(os.path.join(d, pre_r + t + '.csv')) for t in 'abcdefg'}
        old = load(os.path.join(d, 'r_old.csv')) if tag == 'bs=16' else None
        if not any(xs.values()) and not any(rs.values()):
            continue
        print('\n%s  (ms per layer; * = fastest forced tile)' % tag)
        print('layer k s cin cout | default | ' + ' '.join('%7s' % t for t in 'ABCDE') + ' | ' + ' '.join('%7s' % t for t in 'abcdefg') + ' | old1x1')
        seen = set()
        tot_def = tot_best = 0.0
        for i, r in enumerate(base):
            key = (r['k'], r['stride'], r['cin'], r['cout'])
            ms = float(r['ms'])
            cand = {}
            if r['k'] == '3':
                cand = {t: float(v[i]['ms']) for t, v in xs.items() if v}
            else:
                cand = {t: float(v[i]['ms']) for t, v in rs.items() if v}
            best = min(cand.values()) if cand else ms
            tot_def += ms
            tot_best += min(best, ms)
            if key in seen:
                continue
            seen.add(key)
            def cell(t, group):
                v = group.get(t)
                if not v:
                    return '      -'
                x = float(v[i]['ms'])
                return '%6.4f%s' % (x, '*' if cand and x == best else ' ')
            print('%3d %s %s %4s %4s | %.4f | %s | %s | %s' % (
                i, r['k'], r['stride'], r['cin'], r['cout'], ms,
                ' '.join(cell(t, xs) if r['k'] == '3' else '      -' for t in 'ABCDE'),
                ' '.join(cell(t, rs) if r['k'] == '1' else '      -' for t in 'abcdefg'),
                ('%.4f' % float(old[i]['ms'])) if old and r['k'] == '1' else '-'))
        print('sum of layers: default %.3f ms, best forced tile per layer %.3f ms' % (tot_def, tot_best))


if __name__ == '__main__':
    main()
