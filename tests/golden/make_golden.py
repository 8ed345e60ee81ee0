"""Generate golden vectors by IMPORTING the reference's numpy functions unmodified from /root/reference
(under empty stub modules named `tensorflow` and `cv2`, which those functions never touch).

    python tests/golden/make_golden.py          # writes tests/golden/*.npz (committed)

Only runs where /root/reference exists (the build container); the GPU box and the tests consume the
committed .npz files.  What is pinned here: py_nms, cpu_nms (utils/nms_utils.py:51-123), process_box
(utils/data_utils.py:51-115), parse_anchors (utils/misc_utils.py:31-37).  Inputs are tie-free in score
(the reference's argsort is an unstable quicksort, so ties are unspecified there).
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    tf = types.ModuleType('tensorflow')
    tf.contrib = types.SimpleNamespace(slim=None)
    core = types.ModuleType('tensorflow.core')
    fwk = types.ModuleType('tensorflow.core.framework')
    spb = types.ModuleType('tensorflow.core.framework.summary_pb2')
    fwk.summary_pb2 = spb
    core.framework = fwk
    tf.core = core
    sys.modules.update({'tensorflow': tf, 'tensorflow.core': core, 'tensorflow.core.framework': fwk,
                        'tensorflow.core.framework.summary_pb2': spb, 'cv2': types.ModuleType('cv2')})
    sys.path.insert(0, REF)
    from utils import nms_utils, data_utils, misc_utils   # noqa: the reference's own modules
    return nms_utils, data_utils, misc_utils


def unique_scores(rng, shape):
    """fp32 scores in (0,1) with no duplicates (so the reference's unstable sort is well defined)."""
    n = int(np.prod(shape))
    s = rng.permutation(n).astype(np.float64)
    s = (s + rng.uniform(0.1, 0.9, n)) / (n + 1)
    s = s.astype(np.float32)
    assert len(np.unique(s)) == n
    return s.reshape(shape)


def make_boxes(rng, n, size=416.0, wmin=8.0, wmax=256.0):
    c = rng.uniform(0, size, (n, 2))
    wh = rng.uniform(wmin, wmax, (n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)


def main():
    nms_utils, data_utils, misc_utils = import_reference()
    out = {}

    # ---- py_nms: three sizes, two thresholds ------------------------------------------------------
    cases = []
    for ci, (seed, n, maxb, thr) in enumerate([(10, 40, 50, 0.5), (11, 300, 50, 0.45), (12, 300, 20, 0.3),
                                                 (13, 1200, 200, 0.45), (14, 64, 50, 0.0)]):
        rng = np.random.RandomState(seed)
        boxes = make_boxes(rng, n)
        if ci == 4:   # degenerate + tiny boxes exercise the +1 arithmetic (ovr > 1, negative denominators)
            boxes[:16, 2:] = boxes[:16, :2] + rng.uniform(0, 0.5, (16, 2)).astype(np.float32)
            boxes[16:24, 2:] = boxes[16:24, :2]
        scores = unique_scores(rng, (n,))
        keep = nms_utils.py_nms(boxes, scores, max_boxes=maxb, iou_thresh=thr)
        cases.append(dict(boxes=boxes, scores=scores, max_boxes=maxb, iou_thresh=thr,
                          keep=np.asarray(keep, np.int64)))
    out['py_nms_n'] = np.int64(len(cases))
    for i, c in enumerate(cases):
        for k, v in c.items():
            out['py_nms_%d_%s' % (i, k)] = np.asarray(v)

    # ---- cpu_nms: multi-class ---------------------------------------------------------------------
    cases = []
    for seed, n, C, maxb, sthr, ithr in [(20, 500, 6, 30, 0.3, 0.45), (21, 2000, 20, 50, 0.5, 0.5),
                                         (22, 200, 4, 10, 0.99999, 0.5)]:
        rng = np.random.RandomState(seed)
        boxes = make_boxes(rng, n)
        scores = unique_scores(rng, (n, C))
        b, s, l = nms_utils.cpu_nms(boxes[None], scores[None], C, max_boxes=maxb, score_thresh=sthr,
                                    iou_thresh=ithr)
        none = b is None
        cases.append(dict(boxes=boxes, scores=scores, num_classes=C, max_boxes=maxb, score_thresh=sthr,
                          iou_thresh=ithr, is_none=none,
                          out_boxes=np.zeros((0, 4), np.float32) if none else b,
                          out_scores=np.zeros((0,), np.float32) if none else s,
                          out_labels=np.zeros((0,), np.int32) if none else l))
    out['cpu_nms_n'] = np.int64(len(cases))
    for i, c in enumerate(cases):
        for k, v in c.items():
            out['cpu_nms_%d_%s' % (i, k)] = np.asarray(v)

    # ---- parse_anchors ----------------------------------------------------------------------------
    anchors = misc_utils.parse_anchors(os.path.join(REF, 'data', 'yolo_anchors.txt'))
    out['anchors'] = anchors

    # ---- process_box: y_true layout (utils/data_utils.py:51-115) ----------------------------------
    rng = np.random.RandomState(30)
    pb = []
    for i in range(3):
        K = rng.randint(1, 11)
        wh = rng.uniform(10, 300, (K, 2))
        c = np.stack([rng.uniform(wh[:, 0] / 2, 416 - wh[:, 0] / 2), rng.uniform(wh[:, 1] / 2, 416 - wh[:, 1] / 2)], 1)
        boxes = np.concatenate([c - wh / 2, c + wh / 2, np.ones((K, 1))], axis=1).astype(np.float32)
        labels = rng.randint(0, 80, K).astype(np.int64)
        y13, y26, y52 = data_utils.process_box(boxes, labels, [416, 416], 80, anchors)
        pb.append((boxes, labels, y13, y26, y52))
    out['process_box_n'] = np.int64(len(pb))
    for i, (boxes, labels, y13, y26, y52) in enumerate(pb):
        out['process_box_%d_boxes' % i] = boxes
        out['process_box_%d_labels' % i] = labels
        # store sparsely: indices + values of the non-default entries would be smaller, but the dense
        # arrays compress to a few KB
        out['process_box_%d_y13' % i] = y13
        out['process_box_%d_y26' % i] = y26
        out['process_box_%d_y52' % i] = y52
    np.savez_compressed(os.path.join(OUT, 'reference_numpy_goldens.npz'), **out)
    print('wrote', os.path.join(OUT, 'reference_numpy_goldens.npz'), len(out), 'arrays')


if __name__ == '__main__':
    main()
