#!/usr/bin/env python
"""Driver for per-LAYER memory-side counters of the forward (run under `rocprofv3 --pmc ...`, see tools/pmc_layers.sh):
one calibration copy of a known byte count (1 GiB read + 1 GiB written by a 16-B/lane streaming kernel), one warm-up
forward, then ONE measured forward at the bench configuration (bs=32, 416x416, f32_wino; `c5` as the first argument:
bs=16, 608x608, bf16 storage).  tools/pmc_layers_summary.py maps the conv dispatches of the last forward to layers by
dispatch order.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import yolov3_tensorflow_amd as y3
    import bench
    c5 = len(sys.argv) > 1 and sys.argv[1] == 'c5'
    model = y3.yolov3(80, bench.ANCHORS)
    model.compute_dtype = 'bf16' if c5 else 'f32_wino'
    x = torch.rand((16, 608, 608, 3) if c5 else (32, 416, 416, 3), device='cuda')
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
        bench.random_init(1)
        model.forward(x)
        torch.cuda.synchronize()
        a = torch.rand(1 << 28, device='cuda')            # 1 GiB
        b = torch.empty_like(a)
        b.copy_(a)                                        # calibration dispatch (the only 1 GiB device copy of the run)
        torch.cuda.synchronize()
        model.forward(x)
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
