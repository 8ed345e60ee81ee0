#!/usr/bin/env python
"""Timing of the post-processing half of the hot path (decode, NMS) with its rooflines.

    python tools/postproc_bench.py
decode is HBM-bound: algorithmic bytes = read 3.62 MB + write 3.62 MB (+3.41 MB fused scores) per image @416.
NMS is data dependent: timed on (a) a sparse detector-like score matrix and (b) the adversarial dense stress
matrix of tests/test_nms_gpu.py, at the thresholds of the reference's call sites, next to the C oracle on one core.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def timeit(fn, iters=20):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    import torch
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.utils import nms_utils
    from yolov3_tensorflow_amd import _lib
    import bench
    from oracle import nms_ref
    from conftest import make_boxes
    dev = 'cuda'
    model = y3.yolov3(80, bench.ANCHORS)
    for n, size in ((32, 416), (16, 608), (1, 416)):
        model.img_size = [size, size]
        fms = [torch.randn((n, size // s, size // s, 255), device=dev) * 2 for s in (32, 16, 8)]
        B = 3 * sum((size // s) ** 2 for s in (32, 16, 8))
        for ws in (False, True):
            ms = timeit(lambda: model.predict(fms, with_scores=ws))
            byts = n * B * (85 * 4 + (4 + 1 + 80) * 4 + (80 * 4 if ws else 0))
            print('decode  N=%2d @%d scores=%d: %.4f ms  %.2f TB/s (algorithmic %.1f MB)' %
                  (n, size, int(ws), ms, byts / ms / 1e9, byts / 1e6))
    rng = np.random.RandomState(2)
    B, C = 10647, 80
    boxes = make_boxes(rng, B)
    dense = (rng.rand(B, C) * rng.rand(B, C)).astype(np.float32)
    # detector-like: 60 true objects, each producing ~12 overlapping high-score boxes in one class, plus noise
    sparse = (rng.rand(B, C) * 0.008).astype(np.float32)
    for o in range(60):
        c = rng.randint(0, C)
        idx = rng.randint(0, B, 12)
        boxes[idx] = boxes[idx[0]] + rng.uniform(-6, 6, (12, 4)).astype(np.float32)
        sparse[idx, c] = rng.uniform(0.3, 0.98, 12).astype(np.float32)
    for name, scores in (('sparse', sparse), ('dense-stress', dense)):
        for (mb, st, it) in ((200, 0.3, 0.45), (400, 0.01, 0.45)):
            for n in (1, 32):
                b = torch.from_numpy(boxes).to(dev)[None].repeat(n, 1, 1).contiguous()
                s = torch.from_numpy(scores).to(dev)[None].repeat(n, 1, 1).contiguous()
                ms = timeit(lambda: nms_utils._run_nms(_lib.Y3_NMS_TF, b, s, C, mb, st, it), iters=5)
                cand = int((scores >= st).sum())
                line = 'gpu_nms %-12s max_boxes=%d score>=%.2f N=%2d: %.3f ms/call (%.3f ms/image), %d candidates/image' % (
                    name, mb, st, n, ms, ms / n, cand)
                if n == 1:
                    t0 = time.time()
                    out = nms_ref.c_per_class('tf', boxes, scores, C, mb, st, it)
                    line += ' ; C oracle 1 core: %.2f ms (%d detections)' % ((time.time() - t0) * 1e3, len(out[0]))
                print(line, flush=True)


if __name__ == '__main__':
    main()
