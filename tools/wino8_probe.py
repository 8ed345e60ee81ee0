#!/usr/bin/env python
"""Launch time of the Winograd conv on the network's five layer shapes (bs=32), for the library Y3_LIB_PATH points at.
Used with the knock-out builds of tools/wino8_probe.sh (which compute garbage: only the times mean anything)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from yolov3_tensorflow_amd import engine, framework as fw
    dev = fw.default_device()
    out = []
    for (h, cin, cout) in ((13, 512, 1024), (26, 256, 512), (52, 128, 256), (104, 64, 128), (208, 32, 64)):
        n = 32
        x = torch.randn((n, h, h, cin), device=dev)
        w = torch.randn((3, 3, cin, cout), device=dev) * float(np.sqrt(2.0 / (9 * cin)))
        wu = engine.pack_wino(w)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        r = None if os.environ.get('Y3_PROBE_NORES') else torch.randn((n, h, h, cout), device=dev)
        for _ in range(5):
            engine.conv2d_fwd_wino(x, wu, sc, sh, cout, True, residual=r)
        iters = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            engine.conv2d_fwd_wino(x, wu, sc, sh, cout, True, residual=r)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    print(os.environ.get('Y3_TAG', ''), ' '.join('%7.1f' % v for v in out), flush=True)


if __name__ == '__main__':
    main()
