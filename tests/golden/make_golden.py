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


def eval_goldens(nms_utils, data_utils, anchors):
    """Golden vectors of the evaluation bookkeeping (utils/eval_utils.py:13-423, utils/data_utils.py:15-48):
    calc_iou, voc_ap, voc_eval, parse_line, parse_gt_rec, evaluate_on_cpu -> reference_eval_goldens.npz."""
    import io
    import contextlib
    from utils import eval_utils
    if not hasattr(np, 'Inf'):
        np.Inf = np.inf      # the reference spells it np.Inf (removed in NumPy 2.0); same value
    out = {}
    rng = np.random.RandomState(40)
    # ---- calc_iou ---------------------------------------------------------------------------------
    pb, tb = make_boxes(rng, 37).astype(np.float64), make_boxes(rng, 11).astype(np.float64)
    out['iou_pred'], out['iou_true'], out['iou_out'] = pb, tb, eval_utils.calc_iou(pb, tb)
    # ---- voc_ap -----------------------------------------------------------------------------------
    n = 200
    tp = (rng.uniform(size=n) < 0.6).astype(np.float64)
    ctp, cfp = np.cumsum(tp), np.cumsum(1 - tp)
    rec, prec = ctp / 150.0, ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
    out['ap_rec'], out['ap_prec'] = rec, prec
    out['ap_area'] = np.float64(eval_utils.voc_ap(rec, prec, False))
    out['ap_07'] = np.float64(eval_utils.voc_ap(rec, prec, True))
    # ---- annotation file -> parse_line / parse_gt_rec -----------------------------------------------
    lines = []
    n_img, n_cls = 12, 5
    for i in range(n_img):
        w, h = int(rng.randint(200, 900)), int(rng.randint(200, 900))
        k = int(rng.randint(1, 7))
        parts = ['%d' % i, 'img_%d.jpg' % i, '%d' % w, '%d' % h]
        for _ in range(k):
            x0, y0 = rng.uniform(0, w * 0.7), rng.uniform(0, h * 0.7)
            x1, y1 = x0 + rng.uniform(10, w * 0.3), y0 + rng.uniform(10, h * 0.3)
            parts += ['%d' % rng.randint(0, n_cls), '%.2f' % x0, '%.2f' % y0, '%.2f' % x1, '%.2f' % y1]
        lines.append(' '.join(parts))
    out['ann_lines'] = np.array(lines)
    idx, path, boxes, labels, w, h = data_utils.parse_line(lines[3])
    out['pl_idx'], out['pl_path'], out['pl_boxes'], out['pl_labels'], out['pl_wh'] = \
        np.int64(idx), np.array(path), boxes, labels, np.array([w, h], np.int64)
    ann = os.path.join(OUT, '_ann_tmp.txt')
    with open(ann, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    gts = {}
    for lb in (True, False):
        eval_utils.gt_dict = {}
        gd = eval_utils.parse_gt_rec(ann, [416, 416], lb)
        gts[lb] = {k: [list(map(float, o)) for o in v] for k, v in gd.items()}
        flat = np.array([[k] + o for k, v in sorted(gd.items()) for o in v], np.float64)
        out['gt_rec_lb%d' % int(lb)] = flat
    os.remove(ann)
    # ---- voc_eval: detections = jittered ground truth + false positives --------------------------------
    gd = gts[True]
    preds = []
    for img_id, objs in sorted(gd.items()):
        for o in objs:
            if rng.uniform() < 0.8:
                j = rng.normal(0, 6.0, 4)
                preds.append([img_id, o[0] + j[0], o[1] + j[1], o[2] + j[2], o[3] + j[3],
                              float(rng.uniform(0.3, 1.0)), int(o[4])])
            if rng.uniform() < 0.3:    # duplicate detection of the same object
                j = rng.normal(0, 3.0, 4)
                preds.append([img_id, o[0] + j[0], o[1] + j[1], o[2] + j[2], o[3] + j[3],
                              float(rng.uniform(0.3, 1.0)), int(o[4])])
        for _ in range(2):
            x0, y0 = rng.uniform(0, 300, 2)
            preds.append([img_id, x0, y0, x0 + rng.uniform(10, 100), y0 + rng.uniform(10, 100),
                          float(rng.uniform(0.3, 1.0)), int(rng.randint(0, n_cls))])
    out['voc_preds'] = np.array(preds, np.float64)
    res = []
    for c in range(n_cls + 1):     # the last class has neither ground truth nor detections
        with contextlib.redirect_stdout(io.StringIO()):
            for m07 in (False, True):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    res.append([c, int(m07)] + [float(v) for v in
                                                 eval_utils.voc_eval(gd, preds, c, iou_thres=0.5, use_07_metric=m07)])
    out['voc_results'] = np.array(res, np.float64)
    # ---- evaluate_on_cpu: y_true from process_box, y_pred = noisy decode of y_true + clutter -----------
    C = 6
    ytrue = [[], [], []]
    yp_boxes, yp_confs, yp_probs = [], [], []
    for i in range(3):
        K = rng.randint(2, 8)
        wh = rng.uniform(20, 200, (K, 2))
        c = np.stack([rng.uniform(wh[:, 0] / 2, 416 - wh[:, 0] / 2), rng.uniform(wh[:, 1] / 2, 416 - wh[:, 1] / 2)], 1)
        boxes = np.concatenate([c - wh / 2, c + wh / 2, np.ones((K, 1))], axis=1).astype(np.float32)
        labels = rng.randint(0, C, K).astype(np.int64)
        ys = data_utils.process_box(boxes, labels, [416, 416], C, anchors)
        for j in range(3):
            ytrue[j].append(ys[j])
        M = 60
        pb = make_boxes(rng, M)
        pb[:K] = boxes[:, :4] + rng.normal(0, 4.0, (K, 4)).astype(np.float32)
        pb[K:2 * K] = boxes[:, :4] + rng.normal(0, 2.0, (K, 4)).astype(np.float32)    # duplicates
        conf = unique_scores(rng, (M, 1))
        conf[:2 * K] = 0.5 + conf[:2 * K] / 2
        probs = unique_scores(rng, (M, C)) * 0.3
        lab = np.concatenate([labels, labels, rng.randint(0, C, M - 2 * K)])
        lab[K] = (lab[K] + 1) % C                                                     # a wrong-class hit
        probs[np.arange(M), lab] = 0.7 + probs[np.arange(M), lab]
        yp_boxes.append(pb), yp_confs.append(conf), yp_probs.append(probs)
    y_true = [np.stack(t) for t in ytrue]
    y_pred = [np.stack(yp_boxes), np.stack(yp_confs), np.stack(yp_probs)]
    for j in range(3):
        out['ev_ytrue%d' % j] = y_true[j]
    out['ev_pred_boxes'], out['ev_pred_confs'], out['ev_pred_probs'] = y_pred
    rec, prec = eval_utils.evaluate_on_cpu(y_pred, y_true, C, calc_now=True, max_boxes=50, score_thresh=0.3,
                                           iou_thresh=0.5)
    tpd, tld, pld = eval_utils.evaluate_on_cpu(y_pred, y_true, C, calc_now=False, max_boxes=50, score_thresh=0.3,
                                               iou_thresh=0.5)
    out['ev_recall_precision'] = np.array([rec, prec], np.float64)
    out['ev_dicts'] = np.array([[tpd[c], tld[c], pld[c]] for c in range(C)], np.int64)
    # ---- plot_utils.get_color_table (utils/plot_utils.py:9-14) ------------------------------------------
    from utils import plot_utils
    table = plot_utils.get_color_table(80)
    out['color_table_80'] = np.array([table[i] for i in range(80)], np.int64)
    np.savez_compressed(os.path.join(OUT, 'reference_eval_goldens.npz'), **out)
    print('wrote', os.path.join(OUT, 'reference_eval_goldens.npz'), len(out), 'arrays')


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
    eval_goldens(nms_utils, data_utils, anchors)


if __name__ == '__main__':
    main()
