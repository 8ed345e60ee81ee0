#!/usr/bin/env python
"""Phase timing of conv_wino44v_f32_kernel from a -DW44V_PROBE build (tools/build_variant.py probe y3_conv_wino44.hip -DW44V_PROBE;
Y3_LIB_PATH=tools/_probe/lib_probe.so): per workgroup the 100 MHz clock at start / after the prologue / after the K-loop /
after the staging barrier of the tail / at the end, and where it ran.

    python tools/wino44v_probe.py [grid cin cout [bs]]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from yolov3_tensorflow_amd import engine, _lib
    g, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (52, 128, 256)
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    L = _lib.lib()
    fn = L.y3_debug_w44v_probe
    fn.argtypes = [ctypes.c_void_p]
    x = torch.rand((n, g, g, cin), device='cuda')
    w = torch.randn((3, 3, cin, cout), device='cuda') * 0.05
    r = torch.rand((n, g, g, cout), device='cuda')
    sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
    w4 = engine.pack_wino44(w)
    for _ in range(3):
        engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, residual=r)
    buf = torch.zeros((8192, 8), dtype=torch.int64, device='cuda')
    assert fn(buf.data_ptr()) == 0
    engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, residual=r)
    torch.cuda.synchronize()
    fn(None)
    b = buf.cpu().numpy()
    live = b[:, 3] > 0
    b = b[live]
    t0 = b[:, 0].min()
    t = (b[:, :5] - t0) / 100.0                     # us
    hw, xcc = b[:, 7] & 0xffffffff, b[:, 7] >> 32
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)
    print('%d workgroups with work; kernel span %.1f us' % (len(b), t[:, 3].max()))
    for name, d in (('prologue', t[:, 1] - t[:, 0]), ('K-loop', t[:, 2] - t[:, 1]), ('tail: transform + staging', t[:, 4] - t[:, 2]),
                    ('tail: store phase', t[:, 3] - t[:, 4]), ('whole block', t[:, 3] - t[:, 0])):
        print('  %-28s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us' % (name, d.mean(), np.percentile(d, 10),
              np.percentile(d, 50), np.percentile(d, 90), d.max()))
    # start-time histogram: when do blocks start (rounds?)
    hist, edges = np.histogram(t[:, 0], bins=16)
    print('  block starts per %.1f us bin: %s' % (edges[1] - edges[0], ' '.join(str(h) for h in hist)))
    # one CU's timeline
    for c in np.unique(cu)[:3]:
        rows = t[cu == c]
        rows = rows[np.argsort(rows[:, 0])]
        print('  CU %06x:' % c)
        for rr in rows:
            print('     start %6.2f  prologue-> %6.2f  K-loop-> %6.2f  staged-> %6.2f  end %6.2f' % (rr[0], rr[1], rr[2], rr[4], rr[3]))
    # fraction of CU time with k workgroups in their K-loop
    span = t[:, 3].max()
    grid = np.linspace(0, span, 2000)
    ink = np.zeros((len(np.unique(cu)), len(grid)))
    for ci, c in enumerate(np.unique(cu)):
        for rr in t[cu == c]:
            ink[ci] += (grid >= rr[1]) & (grid < rr[2])
    for k in (0, 1, 2):
        print('  CU-time with %d workgroup(s) in the K-loop: %.1f %%' % (k, 100.0 * (ink == k).mean()))


if __name__ == '__main__':
    main()
