# coding: utf-8
"""Mirror of the reference's utils/nms_utils.py: gpu_nms, py_nms, cpu_nms — same names, argument order,
defaults and return conventions — each executed by the HIP NMS kernels (y3_nms) in the matching mode:

  gpu_nms -> Y3_NMS_TF  (tf.image.non_max_suppression semantics, utils/nms_utils.py:8-48)
  py_nms / cpu_nms -> Y3_NMS_PY (the in-tree numpy arithmetic, utils/nms_utils.py:51-123)

Return conventions preserved: gpu_nms returns empty [0,4]/[0]/[0] results when nothing survives,
cpu_nms returns (None, None, None) (utils/nms_utils.py:116-117); labels are int32.
"""
from __future__ import division, print_function

import ctypes

import numpy as np
import torch

from .. import _lib
from .. import framework as fw


_WS_CACHE = {}      # (device index, stream) -> uint8 tensor: y3_nms's scratch, reused by every call on that stream


def _workspace(dev, nbytes):
    """The library's scratch for one y3_nms call.  One buffer per (device, stream), grown on demand: calls on a stream are
    ordered, so the next call may overwrite what the previous one used (the results live in tensors of their own)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS_CACHE[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return ws


class LazyDetections(object):
    """What gpu_nms_batched(lazy=True) / yolov3.detect return: behaves like the list of per-image (boxes, scores, labels[,
    index]) tuples, but the per-image counts travel to the host asynchronously (pinned buffer + event) and the list is built
    on first access - so a caller can enqueue the next batch's forward before it reads this batch's detections, and no
    host round trip sits between two device batches (eval.py:114-123 did one per image)."""

    def __init__(self, tensors, counts_host, event, device, return_index):
        self._t, self._cnt, self._ev, self._dev, self._ri = tensors, counts_host, event, device, return_index
        self._items = None

    def _materialise(self):
        if self._items is None:
            self._ev.synchronize()
            fw.check_context(self._dev)     # surfaces a device-side failure of the forward, if any (stream idle or not)
            ob, osc, ol, oi = self._t
            out = []
            for i, k in enumerate(self._cnt.tolist()):
                item = (ob[i, :k], osc[i, :k], ol[i, :k])
                if self._ri:
                    item = item + (oi[i, :k],)
                out.append(item)
            self._items, self._t = out, None
        return self._items

    def __len__(self):
        return int(self._cnt.shape[0])

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())


def _run_nms(mode, boxes, scores, num_classes, max_boxes, score_thresh, iou_thresh):
    """boxes [n,B,4], scores [n,B,C] device tensors -> (out_boxes, out_scores, out_labels, out_index,
    counts) with per-image capacity C*max_boxes."""
    n, B, C = int(scores.shape[0]), int(scores.shape[1]), int(num_classes)
    max_boxes = int(max_boxes)
    dev = boxes.device
    L = _lib.lib()
    cap = C * max_boxes
    wsb = L.y3_nms_workspace_bytes(n, B, C, max_boxes)
    if wsb == 0:
        raise ValueError("nms: non-positive dimension (n=%d, boxes=%d, classes=%d, max_boxes=%d)" %
                         (n, B, C, max_boxes))
    ws = _workspace(dev, wsb)
    ob = torch.empty((n, cap, 4), dtype=torch.float32, device=dev)
    osc = torch.empty((n, cap), dtype=torch.float32, device=dev)
    ol = torch.empty((n, cap), dtype=torch.int32, device=dev)
    oi = torch.empty((n, cap), dtype=torch.int32, device=dev)
    cnt = torch.empty((n,), dtype=torch.int32, device=dev)
    _lib.check(L.y3_nms(fw.context(dev), mode, fw.ptr(boxes), fw.ptr(scores), n, B, C, max_boxes,
                        ctypes.c_float(score_thresh), ctypes.c_float(iou_thresh), fw.ptr(ws),
                        ctypes.c_size_t(wsb), fw.ptr(ob), fw.ptr(osc), fw.ptr(ol), fw.ptr(oi), fw.ptr(cnt)))
    return ob, osc, ol, oi, cnt


def gpu_nms(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, nms_thresh=0.5):
    """
    Perform NMS on the GPU (reference utils/nms_utils.py:8-48; single image).

    params:
        boxes: tensor of shape [1, 10647, 4] # 10647=(13*13+26*26+52*52)*3, for input 416*416 image
        scores: tensor of shape [1, 10647, num_classes], score=conf*prob
        num_classes: total number of classes
        max_boxes: integer, maximum number of predicted boxes you'd like PER CLASS, default is 50
        score_thresh: boxes with score < score_thresh are dropped (score >= thresh is kept)
        nms_thresh: real value, "intersection over union" threshold used for NMS filtering
    returns boxes [K,4], score [K], label [K] (int32) device tensors, classes ascending.
    """
    b = fw.as_device_f32(boxes).reshape(1, -1, 4)       # '-1': we do nms for a single image
    s = fw.as_device_f32(scores).reshape(1, -1, num_classes)
    ob, osc, ol, _, cnt = _run_nms(_lib.Y3_NMS_TF, b, s, num_classes, max_boxes, score_thresh, nms_thresh)
    k = int(cnt[0].item())
    fw.check_context(b.device)      # the stream is idle here: surface a device-side failure of the forward, if any
    return ob[0, :k], osc[0, :k], ol[0, :k]


def gpu_nms_batched(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, nms_thresh=0.5,
                    return_index=False, lazy=False):
    """Extension: the same op over a batch [N,B,4]/[N,B,C] in one launch set; returns per-image lists.
    lazy=True: a LazyDetections (same indexing / iteration; the host waits for the counts on first access only)."""
    b = fw.as_device_f32(boxes)
    s = fw.as_device_f32(scores)
    ob, osc, ol, oi, cnt = _run_nms(_lib.Y3_NMS_TF, b, s, num_classes, max_boxes, score_thresh, nms_thresh)
    if lazy:
        cnt_h = torch.empty(cnt.shape, dtype=cnt.dtype, pin_memory=True)
        cnt_h.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(b.device))
        return LazyDetections((ob, osc, ol, oi), cnt_h, ev, b.device, return_index)
    cnt_h = cnt.cpu().tolist()
    fw.check_context(b.device)      # the stream is idle here: surface a device-side failure of the forward, if any
    out = []
    for i, k in enumerate(cnt_h):
        item = (ob[i, :k], osc[i, :k], ol[i, :k])
        if return_index:
            item = item + (oi[i, :k],)
        out.append(item)
    return out


def py_nms(boxes, scores, max_boxes=50, iou_thresh=0.5):
    """
    NMS baseline arithmetic of reference utils/nms_utils.py:51-88, on the GPU.

    Arguments: boxes: shape of [-1, 4]
               scores: shape of [-1,]
               max_boxes: maximum of boxes to be selected
               iou_thresh: iou threshold (a box survives while ovr <= iou_thresh)
    returns the list of kept indices, best score first.
    """
    boxes_np_shape = tuple(boxes.shape)
    assert boxes_np_shape[1] == 4 and len(scores.shape) == 1
    if boxes_np_shape[0] == 0:
        return []
    b = fw.as_device_f32(boxes).reshape(1, -1, 4)
    s = fw.as_device_f32(scores).reshape(1, -1, 1)
    _, _, _, oi, cnt = _run_nms(_lib.Y3_NMS_PY, b, s, 1, max_boxes, float('-inf'), iou_thresh)
    k = int(cnt[0].item())
    return oi[0, :k].cpu().numpy().astype(np.int64).tolist()


def cpu_nms(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, iou_thresh=0.5):
    """
    reference utils/nms_utils.py:91-123 (`cpu_nms`): per-class py_nms; numpy in, numpy out.
    Arguments:
        boxes: shape [1, 10647, 4]
        scores: shape [1, 10647, num_classes]
    returns (boxes [K,4], score [K], label [K] int32) numpy arrays or (None, None, None).
    """
    b = fw.as_device_f32(boxes).reshape(1, -1, 4)
    s = fw.as_device_f32(scores).reshape(1, -1, num_classes)
    ob, osc, ol, _, cnt = _run_nms(_lib.Y3_NMS_PY, b, s, num_classes, max_boxes, score_thresh, iou_thresh)
    k = int(cnt[0].item())
    if k == 0:
        return None, None, None
    return ob[0, :k].cpu().numpy(), osc[0, :k].cpu().numpy(), ol[0, :k].cpu().numpy()
