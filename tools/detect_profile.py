#!/usr/bin/env python
"""The post-processing half of yolov3.detect in the bench's 'detector' score regime (objectness biases -4.6, class biases
-3: bench.run_detect) at eval.py's parameters (400 / 0.01 / 0.45): per-stage wall time of decode + NMS on resident feature
maps, and - under `rocprofv3 --kernel-trace --stats` - the kernels' share.

    python tools/detect_profile.py [max_boxes score_thresh]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import yolov3_tensorflow_amd as y3
    import bench
    from yolov3_tensorflow_amd.utils import nms_utils
    max_boxes = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    score_t = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
    model = y3.yolov3(80, bench.ANCHORS)
    model.compute_dtype = 'f32_wino'
    x = torch.rand((32, 416, 416, 3), device='cuda', generator=torch.Generator(device='cuda').manual_seed(100))
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
        bench.random_init(seed=1)
        for v in [v for v in y3.global_variables(scope='yolov3/yolov3_head') if v.op_name.endswith('/biases')]:
            t = v.tensor.clone().view(3, 85)
            t[:, 4] -= 4.6
            t[:, 5:] -= 3.0
            v.assign(t.view(-1))
        fms = model.forward(x, False)
        torch.cuda.synchronize()

        def sync_time(fn, iters=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3, r
        t_pred, (boxes, _, _, scores) = sync_time(lambda: model.predict(fms, with_scores=True))
        t_nms, dets = sync_time(lambda: nms_utils.gpu_nms_batched(boxes, scores, 80, max_boxes, score_t, 0.45))
        t_raw, _ = sync_time(lambda: nms_utils._run_nms(1, boxes, scores, 80, max_boxes, score_t, 0.45))
        t_det, _ = sync_time(lambda: model.detect(x, max_boxes, score_t, 0.45))
        t_fwd, _ = sync_time(lambda: model.forward(x, False))
    kc = (scores >= score_t).sum(1).flatten().cpu().numpy()          # candidates per (image, class)
    import numpy as np
    sel = np.zeros((32, 80), np.int64)
    for i, d in enumerate(dets):
        sel[i] = np.bincount(d[2].cpu().numpy(), minlength=80)
    sel = sel.flatten()
    order = np.argsort(-kc)[:8]
    print('candidates per (image, class): median %d, p90 %d, p99 %d, max %d; classes with K > 512: %d, > 2048: %d, > 8192: %d of %d'
          % (np.median(kc), np.percentile(kc, 90), np.percentile(kc, 99), kc.max(), (kc > 512).sum(), (kc > 2048).sum(), (kc > 8192).sum(), kc.size))
    print('heaviest classes (K, selected): %s; sum over classes of K * selected = %.3g pair tests'
          % ([(int(kc[o]), int(sel[o])) for o in order], float((kc.astype(np.float64) * sel).sum())))
    cand = int((scores >= score_t).sum().item())
    print('detector regime, max_boxes %d, score >= %g: %d candidates / image, %.0f detections / image'
          % (max_boxes, score_t, cand // 32, sum(int(d[0].shape[0]) for d in dets) / 32.0))
    print('predict %.3f ms | gpu_nms_batched %.3f ms (y3_nms + empties alone: %.3f ms) | detect %.3f ms | forward %.3f ms -> '
          'decode + nms = %.3f ms' % (t_pred, t_nms, t_raw, t_det, t_fwd, t_det - t_fwd))


if __name__ == '__main__':
    main()
