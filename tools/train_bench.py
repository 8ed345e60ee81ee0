#!/usr/bin/env python
"""Throughput of one training step (BASELINE configs[3]: 416x416, bs=64 per GPU, SGD) on synthetic batches.

    python tools/train_bench.py [--batch 64] [--steps 5] [--optimizer sgd] [--head-only]
"""
import argparse
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_y_true(n, size, class_num, anchors, seed, device):
    """Device-side stand-in for process_box output: a few object cells per image per scale."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    out = []
    for s, a0 in ((32, 6), (16, 3), (8, 0)):
        gsz = size // s
        y = torch.zeros((n, gsz, gsz, 3, 6 + class_num))
        y[..., -1] = 1.0
        for i in range(n):
            for _ in range(3):
                cy, cx, k = (int(torch.randint(0, gsz, (1,), generator=g)), int(torch.randint(0, gsz, (1,), generator=g)),
                             int(torch.randint(0, 3, (1,), generator=g)))
                w, h = float(anchors[a0 + k][0]), float(anchors[a0 + k][1])
                y[i, cy, cx, k, 0:4] = torch.tensor([(cx + 0.5) * s, (cy + 0.5) * s, w, h])
                y[i, cy, cx, k, 4] = 1.0
                y[i, cy, cx, k, 5 + int(torch.randint(0, class_num, (1,), generator=g))] = 1.0
        out.append(y.to(device))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--optimizer', default='sgd')
    ap.add_argument('--head-only', action='store_true')
    ap.add_argument('--precision', default='f32', help='f32 | f32_bf16x6 (forward + stride-1 dgrad on the bf16 pipe)')
    a = ap.parse_args()
    import torch
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    import bench
    model = y3.yolov3(80, bench.ANCHORS, batch_norm_decay=0.99)
    model.compute_dtype = a.precision
    x = torch.rand((a.batch, a.size, a.size, 3), device='cuda')
    yt = synthetic_y_true(a.batch, a.size, 80, bench.ANCHORS, 0, 'cuda')
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
        bench.random_init(1)
        upd = None
        if a.head_only:
            upd = [v for v in y3.global_variables(scope='yolov3') if v.op_name.startswith('yolov3/yolov3_head')]
        trainer = training.Trainer(model, config_optimizer(a.optimizer, 1e-4), update_vars=upd)
        for _ in range(2):
            loss = trainer.step(x, yt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = trainer.step(x, yt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    flops = 3 * bench.conv_flops(model._train['topo_table'] if 'topo_table' in model._train else
                                 [(l['k'], l['stride'], l['cin'], l['cout'], l['bn']) for l in model._train['topo'].layers],
                                 a.batch, a.size, a.size).sum()
    print('train step [%s]: batch %d @%d, %s%s: %.1f ms/step, %.1f images/s, %.1f TFLOP/s (3x forward FLOPs), loss %.3f, peak mem %.1f GB'
          % (a.precision, a.batch, a.size, a.optimizer, ' head-only' if a.head_only else '', dt * 1e3, a.batch / dt,
             flops / dt / 1e12, float(loss[0]), torch.cuda.max_memory_allocated() / 1e9))


if __name__ == '__main__':
    main()
