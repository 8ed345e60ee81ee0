#!/usr/bin/env python
"""F(4x4,3x3) kernel (y3_conv2d_fwd_wino44) against the F(2x2,3x3) eight-wave kernel (y3_conv2d_fwd_wino) on the five
stride-1 3x3 layer shapes of the bs=32 416x416 forward, back to back on random tensors (us per launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    from yolov3_tensorflow_amd import engine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    print('shape (bs=%d)            F(2x2) us   F(4x4) us   ratio   F(4x4) TF/s issued (of 157.3)   F(4x4) persistent us' % n)
    for g, cin, cout in ((208, 32, 64), (104, 64, 128), (52, 128, 256), (26, 256, 512), (13, 512, 1024)):
        x = torch.rand((n, g, g, cin), device='cuda')
        w = torch.randn((3, 3, cin, cout), device='cuda') * 0.05
        sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
        w2, w4 = engine.pack_wino(w), engine.pack_wino44(w)
        t2 = timed(lambda: engine.conv2d_fwd_wino(x, w2, sc, sh, cout, True))
        t4 = timed(lambda: engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, use_workspace=False))
        t4p = timed(lambda: engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True))
        tiles = n * ((g + 3) // 4) ** 2
        issued = tiles * 36 * cin * cout * 2 / (t4 * 1e-6) / 1e12
        print('%3dx%-3d %4d->%-4d       %8.1f    %8.1f    %.2f    %.1f    %8.1f' % (g, g, cin, cout, t2, t4, t2 / t4, issued, t4p), flush=True)


if __name__ == '__main__':
    main()
