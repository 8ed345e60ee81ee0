"""CPU restatement of the NMS definitions (TEST INFRASTRUCTURE — see oracle/__init__.py).

Two implementations of the same semantics:
  * pure Python/numpy-fp32 functions below (readable, small cases);
  * oracle/nms_ref.c via ctypes (`c_*` functions; full-size cases, bench cpu_baseline).
tf_* = tf.image.non_max_suppression as used by utils/nms_utils.py:8-48 (parity unpinned: third-party op);
py_* = utils/nms_utils.py:51-123 (pinned to the reference's own functions through tests/golden/).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_f32 = np.float32


def _order(scores):
    """(score descending, index ascending)."""
    return np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))


def tf_iou(bi, bj):
    bi = bi.astype(_f32); bj = bj.astype(_f32)
    ymin_i, xmin_i = min(bi[0], bi[2]), min(bi[1], bi[3])
    ymax_i, xmax_i = max(bi[0], bi[2]), max(bi[1], bi[3])
    ymin_j, xmin_j = min(bj[0], bj[2]), min(bj[1], bj[3])
    ymax_j, xmax_j = max(bj[0], bj[2]), max(bj[1], bj[3])
    area_i = _f32(ymax_i - ymin_i) * _f32(xmax_i - xmin_i)
    area_j = _f32(ymax_j - ymin_j) * _f32(xmax_j - xmin_j)
    if area_i <= 0 or area_j <= 0:
        return _f32(0)
    iy = max(_f32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), _f32(0))
    ix = max(_f32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), _f32(0))
    inter = _f32(iy * ix)
    return _f32(inter / _f32(_f32(area_i + area_j) - inter))


def tf_nms(boxes, scores, max_output_size, iou_threshold):
    boxes = np.asarray(boxes, _f32); scores = np.asarray(scores, _f32)
    selected = []
    for i in _order(scores):
        if len(selected) >= max_output_size:
            break
        keep = True
        for j in reversed(selected):
            if tf_iou(boxes[i], boxes[j]) > _f32(iou_threshold):
                keep = False
                break
        if keep:
            selected.append(int(i))
    return selected


def py_nms(boxes, scores, max_boxes=50, iou_thresh=0.5):
    """utils/nms_utils.py:51-88 with the tie order fixed to (score desc, index asc)."""
    boxes = np.asarray(boxes, _f32); scores = np.asarray(scores, _f32)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = _order(scores)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1 = np.maximum(x1[i], x1[order[1:]]); yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]]); yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(_f32(0.0), xx2 - xx1 + _f32(1)); h = np.maximum(_f32(0.0), yy2 - yy1 + _f32(1))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= _f32(iou_thresh))[0]
        order = order[inds + 1]
    return keep[:max_boxes]


def per_class(mode, boxes, scores, num_classes, max_boxes, score_thresh, iou_thresh):
    """gpu_nms (mode 'tf', utils/nms_utils.py:8-48) / cpu_nms (mode 'py', :91-123) driver.
    Returns (boxes [K,4], scores [K], labels [K] int32, index [K] int32)."""
    boxes = np.asarray(boxes, _f32).reshape(-1, 4)
    scores = np.asarray(scores, _f32).reshape(-1, num_classes)
    ob, os_, ol, oi = [], [], [], []
    for c in range(num_classes):
        idx = np.where(scores[:, c] >= _f32(score_thresh))[0]
        if len(idx) == 0:
            continue
        fb, fs = boxes[idx], scores[idx, c]
        sel = tf_nms(fb, fs, max_boxes, iou_thresh) if mode == 'tf' else py_nms(fb, fs, max_boxes, iou_thresh)
        ob.append(fb[sel]); os_.append(fs[sel])
        ol.append(np.full(len(sel), c, np.int32)); oi.append(idx[sel].astype(np.int32))
    if not ob:
        return (np.zeros((0, 4), _f32), np.zeros((0,), _f32), np.zeros((0,), np.int32), np.zeros((0,), np.int32))
    return np.concatenate(ob), np.concatenate(os_), np.concatenate(ol), np.concatenate(oi)


# ---- C implementation ------------------------------------------------------------------------------
_clib = None


def build_c(verbose=False):
    subprocess.check_call(['make', '-C', HERE] + ([] if verbose else ['-s']))
    return os.path.join(HERE, '_build', 'liboracle.so')


def clib():
    global _clib
    if _clib is None:
        path = os.path.join(HERE, '_build', 'liboracle.so')
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, 'nms_ref.c')):
            build_c()
        L = ctypes.CDLL(path)
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
        for name in ('tf_nms', 'py_nms'):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ip]
        L.per_class_nms.restype = ctypes.c_int
        L.per_class_nms.argtypes = [ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_float, ctypes.c_float, fp, fp, ip, ip]
        _clib = L
    return _clib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def c_single(mode, boxes, scores, max_boxes, iou_thresh):
    boxes = np.ascontiguousarray(boxes, _f32).reshape(-1, 4); scores = np.ascontiguousarray(scores, _f32)
    sel = np.zeros(max(max_boxes, 1), np.int32)
    fn = clib().tf_nms if mode == 'tf' else clib().py_nms
    n = fn(_fp(boxes), _fp(scores), len(scores), int(max_boxes), ctypes.c_float(iou_thresh), _ip(sel))
    return sel[:n].tolist()


def c_per_class(mode, boxes, scores, num_classes, max_boxes, score_thresh, iou_thresh):
    boxes = np.ascontiguousarray(boxes, _f32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, _f32).reshape(-1, num_classes)
    cap = num_classes * max_boxes
    ob = np.zeros((cap, 4), _f32); os_ = np.zeros(cap, _f32)
    ol = np.zeros(cap, np.int32); oi = np.zeros(cap, np.int32)
    n = clib().per_class_nms(0 if mode == 'tf' else 1, _fp(boxes), _fp(scores), boxes.shape[0], num_classes,
                             int(max_boxes), ctypes.c_float(score_thresh), ctypes.c_float(iou_thresh),
                             _fp(ob), _fp(os_), _ip(ol), _ip(oi))
    return ob[:n].copy(), os_[:n].copy(), ol[:n].copy(), oi[:n].copy()
