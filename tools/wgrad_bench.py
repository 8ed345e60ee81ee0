#!/usr/bin/env python
"""Weight-gradient kernel (y3_conv_wgrad) on the layer shapes of BASELINE configs[3] (bs=64 @416).

    python tools/wgrad_bench.py [--batch 64] [--iters 10]          # env Y3_WGRAD_OLD_SPLIT=1 = the round-1 split rule
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (k, stride, cin, cout, input hw, count in the network)
SHAPES = [(3, 1, 128, 256, 52, 11), (3, 1, 256, 512, 26, 11), (3, 1, 512, 1024, 13, 7), (3, 1, 64, 128, 104, 2),
          (3, 1, 32, 64, 208, 1), (3, 2, 128, 256, 104, 1), (3, 2, 512, 1024, 26, 1), (1, 1, 256, 128, 52, 10),
          (1, 1, 512, 256, 26, 11), (1, 1, 1024, 512, 13, 7), (1, 1, 128, 64, 104, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--wino', action='store_true', help='y3_conv_wgrad_wino for the shapes it takes')
    a = ap.parse_args()
    import torch
    from yolov3_tensorflow_amd import framework as fw, _lib
    L, ctx = _lib.lib(), fw.context()
    dev = fw.default_device()
    tot_ms = tot_flop = 0.0
    for k, s, cin, cout, hw, cnt in SHAPES:
        n = a.batch
        ho = hw // s
        x = torch.randn((n, hw, hw, cin), device=dev)
        dz = torch.randn((n, ho, ho, cout), device=dev)
        d = _lib.ConvDesc(n, hw, hw, cin, 0, cout, k, s, 0)
        dw = torch.empty((k, k, cin, cout), device=dev)
        wino = a.wino and L.y3_conv_wgrad_wino_eligible(ctypes.byref(d)) == 1
        nbytes = (L.y3_conv_wgrad_wino_scratch_bytes if wino else L.y3_conv_wgrad_scratch_bytes)(ctypes.byref(d))
        sc = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        fn = L.y3_conv_wgrad_wino if wino else L.y3_conv_wgrad
        run = lambda: _lib.check(fn(ctx, ctypes.byref(d), fw.ptr(x), fw.ptr(dz), cout, fw.ptr(dw), fw.ptr(sc),
                                    ctypes.c_size_t(sc.numel())))
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        flop = 2.0 * k * k * cin * cout * ho * ho * n
        print('wgrad k%d s%d %4d->%4d @%3d bs%d: %.3f ms  %.1f TF/s  (x%d in the net)%s' % (
            k, s, cin, cout, hw, n, ms, flop / ms / 1e9, cnt, '  [Winograd: TF/s counted as direct-algorithm FLOPs]' if wino else ''))
        tot_ms += ms * cnt
        tot_flop += flop * cnt
    print('weighted: %.2f ms for %d layers, %.1f TF/s [old split %s]' % (
        tot_ms, sum(c[-1] for c in SHAPES), tot_flop / tot_ms / 1e9,
        os.environ.get('Y3_WGRAD_OLD_SPLIT', '0')))


if __name__ == '__main__':
    main()
