#!/usr/bin/env python
"""F(4x4,3x3) conv (y3_conv2d_fwd_wino44) in its two forms - two kernels (input transform written once + batched GEMMs) and one
kernel (transform inside the K-loop) - against the F(2x2,3x3) kernel (y3_conv2d_fwd_wino) on the five stride-1 3x3 layer
shapes of the bs=32 416x416 forward, back to back on random tensors (us per launch)."""
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
    print('shape (bs=%d)            F(2x2) us   F(4x4) 1 kernel us   F(4x4) 2 kernels us   ratio 2k/1k   useful TF/s 2k (of 157.3)' % n)
    shapes = ((208, 32, 64), (104, 64, 128), (52, 128, 256), (26, 256, 512), (13, 512, 1024))
    if os.environ.get('W44_DGRAD') == '1':          # the data-gradient convs of the same layers (channel axes swapped)
        shapes = ((104, 128, 64), (52, 256, 128), (26, 512, 256), (13, 1024, 512))
    for g, cin, cout in shapes:
        x = torch.rand((n, g, g, cin), device='cuda')
        w = torch.randn((3, 3, cin, cout), device='cuda') * 0.05
        r = torch.rand((n, g, g, cout), device='cuda')
        sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
        w2, w4 = engine.pack_wino(w), engine.pack_wino44(w)
        quick = os.environ.get('W44_ONLY2K') == '1'          # (A/B runs of probe builds: only the two-kernel form)
        t2 = 0.0 if quick else timed(lambda: engine.conv2d_fwd_wino(x, w2, sc, sh, cout, True, residual=r))
        t1k = 1.0 if quick else timed(lambda: engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, residual=r, use_workspace=False))
        t2k = timed(lambda: engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, residual=r))
        useful = n * g * g * 2.25 * cin * cout * 2 / (t2k * 1e-6) / 1e12
        print('%3dx%-3d %4d->%-4d       %8.1f    %8.1f    %8.1f    %.2f    %.1f' % (g, g, cin, cout, t2, t1k, t2k, t2k / t1k, useful), flush=True)


if __name__ == '__main__':
    main()
