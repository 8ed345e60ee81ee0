#!/usr/bin/env python
"""A/B of model.inference_streams = 1 / 2 inside one process, alternating: ms per batch for c2 (bs=32, 416, f32_wino) and c5
(bs=16, 608, bf16)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import yolov3_tensorflow_amd as y3
    import bench
    for dtype, size, bs in (('f32_wino', 416, 32), ('bf16', 608, 16)):
        y3.reset_default_graph()
        model = y3.yolov3(80, bench.ANCHORS)
        model.compute_dtype = dtype
        x = torch.rand((bs, size, size, 3), device='cuda')
        with y3.variable_scope('yolov3'):
            model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
            bench.random_init(1)
            counts = [int(v) for v in os.environ.get('Y3_STREAMS_AB', '1,2').split(',')]
            res = {c: [] for c in counts}
            for rep in range(4):
                for ns in counts:
                    model.inference_streams = ns
                    for _ in range(3):
                        model.forward(x)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        model.forward(x)
                    torch.cuda.synchronize()
                    res[ns].append((time.perf_counter() - t0) / 20 * 1e3)
        print('%s %dx%d bs=%d: %s' % (dtype, size, size, bs, '; '.join('%d stream(s) %s ms' % (c, ' '.join('%.3f' % v for v in res[c])) for c in counts)), flush=True)


if __name__ == '__main__':
    main()
