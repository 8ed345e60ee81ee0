#!/usr/bin/env python
"""Per-phase shader-clock cycles of one K-step of the eight-wave Winograd kernel (probe build -DY3_WINO8_CLOCK, built by
tools/wino8_probe_build.sh with EXTRA=-DY3_WINO8_CLOCK): wave 0 (position half 0) and wave 4 (half 1) of workgroup 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from yolov3_tensorflow_amd import engine, framework as fw
    dev = fw.default_device()
    for (h, cin, cout) in ((13, 512, 1024), (26, 256, 512), (52, 128, 256), (104, 64, 128)):
        n = 32
        x = torch.randn((n, h, h, cin), device=dev)
        w = torch.randn((3, 3, cin, cout), device=dev) * float(np.sqrt(2.0 / (9 * cin)))
        wu = engine.pack_wino(w)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        for _ in range(3):
            y = engine.conv2d_fwd_wino(x, wu, sc, sh, cout, True)
        torch.cuda.synchronize()
        c = y.view(-1)[:64].view(torch.int64).cpu().numpy()
        for half in (0, 1):
            o = half * 16
            total, steps = int(c[o]), max(int(c[o + 1]), 1)
            ph = [int(v) / steps for v in c[o + 2: o + 8]]
            segs = max(int(c[o + 8]), 1)
            sg = [int(v) / segs for v in c[o + 9: o + 15]]
            print('H=%3d %4d->%4d half %d: kernel %d cycles, %d K-steps in loops, %.0f cycles per K-step: ' % (h, cin, cout, half, total, steps, sum(ph))
                  + ' | '.join('%.0f' % v for v in ph), flush=True)
            print('      %d segments, cycles per segment: prologue %.0f | K-loop + last K-step %.0f | output transform %.0f | '
                  'stream-K wait %.0f | next-segment prefetch %.0f | finish / publish %.0f' % ((segs,) + tuple(sg)), flush=True)


if __name__ == '__main__':
    main()
