#!/usr/bin/env python
"""One row of F(4x4,3x3) launch times (us, one workgroup per block schedule) on the five stride-1 3x3 shapes of the bs=32
416x416 forward, for the library Y3_LIB_PATH names - the knock-out / variant builds of tools/build_variant.py, all rows in
ONE gpurun call (tools/r04_ko.sh).  Knock-out builds compute wrong results on purpose: timing only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.wino44_bench import timed      # noqa: E402


def main():
    from yolov3_tensorflow_amd import engine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    row = []
    for g, cin, cout in ((208, 32, 64), (104, 64, 128), (52, 128, 256), (26, 256, 512), (13, 512, 1024)):
        x = torch.rand((n, g, g, cin), device='cuda')
        w = torch.randn((3, 3, cin, cout), device='cuda') * 0.05
        sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
        w4 = engine.pack_wino44(w)
        row.append(timed(lambda: engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, use_workspace=False)))
    print('%-28s %s' % (os.path.basename(os.environ.get('Y3_LIB_PATH', 'product')), '  '.join('%7.1f' % t for t in row)), flush=True)


if __name__ == '__main__':
    main()
