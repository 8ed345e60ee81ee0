#!/usr/bin/env python
"""Micro-benchmark of y3_conv2d_fwd on the network's conv shapes (hipEvents on the launch stream).

    Y3_CONV_VARIANT=3 python tools/conv_bench.py [--batch 32] [--iters 20] [--shapes all|main]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (H, k, stride, cin, cout, residual, c_up)
MAIN = [
    (208, 3, 1, 32, 64, True, 0),
    (104, 3, 1, 64, 128, True, 0),
    (52, 3, 1, 128, 256, True, 0),
    (52, 3, 1, 128, 256, False, 0),
    (26, 3, 1, 256, 512, True, 0),
    (13, 3, 1, 512, 1024, True, 0),
    (104, 3, 2, 128, 256, False, 0),
    (52, 1, 1, 256, 128, False, 0),
    (26, 1, 1, 512, 256, False, 0),
    (13, 1, 1, 1024, 512, False, 0),
    (52, 1, 1, 256, 255, False, 0),
    (26, 1, 1, 768, 256, False, 256),
    (208, 1, 1, 64, 32, False, 0),
    (416, 3, 2, 32, 64, False, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--planes', type=int, default=0, help='0: exact fp32 MFMA; 2/3: split-bf16 kernel')
    ap.add_argument('--wino', action='store_true', help='Winograd kernel for the eligible 3x3 s1 shapes')
    ap.add_argument('--only', type=int, default=-1, help='run only this row of the shape table')
    a = ap.parse_args()
    import torch
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    L = _lib.lib()
    var = os.environ.get('Y3_CONV_VARIANT', 'default') + ('/p%d' % a.planes)
    tot = 0.0
    for (h, k, s, cin, cout, resid, c_up) in (MAIN if a.only < 0 else MAIN[a.only:a.only + 1]):
        n = a.batch
        cx = cin - c_up
        x = torch.randn((n, h, h, cx), device=dev)
        xu = torch.randn((n, h // 2, h // 2, c_up), device=dev) if c_up else None
        w = torch.randn((k, k, cin, cout), device=dev) * float(np.sqrt(2.0 / (k * k * cin)))
        if a.planes:
            wp = torch.empty(a.planes * k * k * cout * cin, device=dev, dtype=torch.bfloat16)
            _lib.check(L.y3_pack_conv_weights_split(fw.context(), fw.ptr(w), k, cin, cout, a.planes, fw.ptr(wp)))
        else:
            wp = torch.empty(k * k * cout * cin, device=dev)
            _lib.check(L.y3_pack_conv_weights(fw.context(), fw.ptr(w), k, cin, cout, fw.ptr(wp)))
        sc = torch.ones(cout, device=dev)
        sh = torch.zeros(cout, device=dev)
        r = torch.randn((n, h // s, h // s, cout), device=dev) if resid else None
        run = lambda: engine.conv2d_fwd(x, wp, sc, sh, k, s, cout, True, residual=r, x_up=xu, planes=a.planes)
        if a.wino and engine.wino_eligible(k, s, cin, cout, c_up):
            wu = engine.pack_wino(w)
            run = lambda: engine.conv2d_fwd_wino(x, wu, sc, sh, cout, True, residual=r)
        for _ in range(3):
            y = run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            y = run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * k * k * cin * cout * (h // s) ** 2 * n
        tot += ms
        print('var=%s H=%3d k=%d s=%d %4d->%4d resid=%d up=%3d : %.4f ms  %.1f TF/s' %
              (var, h, k, s, cin, cout, int(resid), c_up, ms, fl / ms / 1e9), flush=True)
    print('var=%s total %.3f ms' % (var, tot))


if __name__ == '__main__':
    main()
