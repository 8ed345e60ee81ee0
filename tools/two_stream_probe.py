#!/usr/bin/env python
"""Does running the bs=32 forward as two concurrent half batches on two HIP streams fill the kernels' partly empty last
rounds?  ms per 32 images: one stream (bs=32) against two streams (bs=16 each, launched alternately)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import yolov3_tensorflow_amd as y3
    import bench
    nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32_wino'
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 416
    bs = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    model = y3.yolov3(80, bench.ANCHORS)
    model.compute_dtype = dtype
    x = torch.rand((bs, size, size, 3), device='cuda')
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    parts = list(x.chunk(nsplit))
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
        bench.random_init(1)
        for _ in range(3):
            model.forward(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model.forward(x)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / 20 * 1e3
        for s, p in zip(streams, parts):
            with torch.cuda.stream(s):
                for _ in range(3):
                    model.forward(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            for s, p in zip(streams, parts):
                with torch.cuda.stream(s):
                    model.forward(p)
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) / 20 * 1e3
        # the half batch alone
        with torch.cuda.stream(streams[0]):
            t0 = time.perf_counter()
            for _ in range(20):
                model.forward(parts[0])
            torch.cuda.synchronize()
            half = (time.perf_counter() - t0) / 20 * 1e3
    print('%s %d: one stream bs=%d: %.3f ms   %d streams bs=%d each: %.3f ms per batch   (one part alone: %.3f ms)'
          % (dtype, size, bs, one, nsplit, bs // nsplit, two, half))


if __name__ == '__main__':
    main()
