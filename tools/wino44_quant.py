#!/usr/bin/env python
"""Block-count quantisation of conv_wino44_f32_kernel: 32x32 images (64 tiles each), Cout = 512 (8 channel blocks), Cin = 256:
blocks = N * 64 / BT * 8.  us per launch against the number of blocks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.wino44_bench import timed


def main():
    from yolov3_tensorflow_amd import engine
    cin, cout = 256, 512
    w = torch.randn((3, 3, cin, cout), device='cuda') * 0.05
    w4 = engine.pack_wino44(w)
    sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
    bt = int(os.environ.get('W44_BT', '16'))
    for n in (4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 64):
        x = torch.rand((n, 32, 32, cin), device='cuda')
        t = timed(lambda: engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, use_workspace=False))
        tp = timed(lambda: engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True))
        blocks = n * 64 // bt * 8
        print('N=%2d blocks=%5d (%.2f per CU)  %7.1f us   %.3f us per block/CU-slot   persistent %7.1f us'
              % (n, blocks, blocks / 256.0, t, t / (blocks / 256.0), tp), flush=True)


if __name__ == '__main__':
    main()
