#!/usr/bin/env python
"""Shader clock and per-K-step cycles of the Winograd kernel (probe build only).

    Y3_EXTRA_HIPCC_FLAGS=-DY3_WINO_CLOCK python -c "import __graft_entry__ as g; g.build()"
    python tools/wino_clock_probe.py

Knock-out variants for the K-step breakdown of DESIGN.md 4.4: add -DY3_WINO_KO=1 (no loads in the K-loop), =2 (no
input transform / LDS writes) or =3 (both); they compute garbage, only the cycle counts mean anything.

The probe build makes workgroup 0 write its s_memtime deltas (whole kernel, K-loops only, K-steps inside those loops)
over the first 24 bytes of the output tensor; hipEvents give the kernel's wall time, the ratio is the shader clock
the kernel actually ran at.
"""
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
        r = torch.randn((n, h, h, cout), device=dev)
        for _ in range(5):
            y = engine.conv2d_fwd_wino(x, wu, sc, sh, cout, True, residual=r)
        iters = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            y = engine.conv2d_fwd_wino(x, wu, sc, sh, cout, True, residual=r)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        c = y.view(-1)[:18].view(torch.int64).cpu().numpy()
        total, loop, steps = int(c[0]), int(c[1]), int(c[2])
        print('H=%3d %4d->%4d: launch %.1f us, workgroup 0: %d cycles -> >= %.2f GHz; '
              'K-loops %d cycles / %d K-steps = %.0f cycles per K-step; outside the K-loops %d cycles'
              % (h, cin, cout, us, total, total / us / 1e3, loop, steps, loop / max(steps, 1), total - loop),
              flush=True)
        print('      phases (cycles, whole kernel): prologue %d | K-loops %d | last K-step %d | output transform %d | '
              'next-block prefetch %d | tail stores %d' % tuple(int(v) for v in c[3:9]), flush=True)


if __name__ == '__main__':
    main()
