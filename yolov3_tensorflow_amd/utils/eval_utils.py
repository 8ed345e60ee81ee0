# coding: utf-8
"""Evaluation bookkeeping of the reference (utils/eval_utils.py) on top of the device-resident detection path —
SURVEY.md §8(f) row 2.  Same function names, argument order and return values as the reference:

  calc_iou            utils/eval_utils.py:13-46     IoU matrix [N,V] (numpy, float64 in / float64 out)
  evaluate_on_cpu     utils/eval_utils.py:49-139    recall / precision of one batch, NMS = cpu_nms semantics
  evaluate_on_gpu     utils/eval_utils.py:142-232   same with the gpu_nms op
  get_preds_gpu       utils/eval_utils.py:235-261   [[image_id, x_min, y_min, x_max, y_max, score, label], ...]
  parse_gt_rec        utils/eval_utils.py:264-307   annotation file -> {img_id: [[x0, y0, x1, y1, label], ...]}
  voc_ap, voc_eval    utils/eval_utils.py:312-423   PASCAL VOC AP (area and 11-point) per class

plus `get_preds_batch`, the MI355X-first form: N images per call, detections produced by `yolov3.detect`
(forward -> decode -> batched per-class NMS on the device, one host transfer per batch instead of two
`sess.run` round trips per image).

The TF plumbing arguments of the reference signatures (`sess`, `pred_boxes_flag`, `pred_scores_flag`) are kept
positionally and ignored; `gpu_nms_op` is a callable `(boxes [1,B,4], scores [1,B,C]) -> (boxes, scores, labels)`
(default: utils.nms_utils.gpu_nms with the arguments bound by the caller).

All arithmetic that decides a match or an AP value is done in float64 numpy in the reference's operation order, so
the results are bit-identical to the reference's (pinned by tests/golden/reference_eval_goldens.npz).
"""
from __future__ import division, print_function

import numpy as np

from .data_utils import parse_line
from .nms_utils import cpu_nms


def _to_numpy(a):
    return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)


def calc_iou(pred_boxes, true_boxes):
    '''
    IoU matrix via numpy broadcasting.
    shape_info: pred_boxes: [N, 4] (x_min, y_min, x_max, y_max)
                true_boxes: [V, 4]
    return: IoU matrix: shape: [N, V]
    '''
    p = np.asarray(pred_boxes)[:, None, :]        # [N, 1, 4]
    t = np.asarray(true_boxes)[None, :, :]        # [1, V, 4]
    wh = np.maximum(np.minimum(p[..., 2:], t[..., 2:]) - np.maximum(p[..., :2], t[..., :2]), 0.)
    inter = wh[..., 0] * wh[..., 1]
    p_wh = p[..., 2:] - p[..., :2]
    t_wh = t[..., 2:] - t[..., :2]
    p_area = p_wh[..., 0] * p_wh[..., 1]
    t_area = t_wh[..., 0] * t_wh[..., 1]
    return inter / (p_area + t_area - inter + 1e-10)


def _ground_truth_of_image(y_true, i):
    """Labels [V] and corner boxes [V,4] of image i, gathered over the three y_true scales in the reference's
    order (scale 13 first, row-major inside a scale)."""
    labels, boxes = [], []
    for j in range(3):
        yt = _to_numpy(y_true[j][i])
        probs = yt[..., 5:-1]
        mask = probs.sum(axis=-1) > 0
        labels.append(np.argmax(probs[mask], axis=-1))
        boxes.append(yt[..., 0:4][mask])
    labels = np.concatenate(labels)
    centre_wh = np.concatenate(boxes).astype(np.float64)   # the reference round-trips through python floats
    corners = np.empty_like(centre_wh)
    corners[:, 0:2] = centre_wh[:, 0:2] - centre_wh[:, 2:4] / 2.
    corners[:, 2:4] = corners[:, 0:2] + centre_wh[:, 2:4]
    return labels, corners


def _evaluate(y_pred, y_true, num_classes, nms_fn, iou_thresh, calc_now):
    """Shared body of evaluate_on_cpu / evaluate_on_gpu.  A ground-truth object counts as found when at least one
    detection has it as its best-IoU object with IoU > iou_thresh and the same label (the reference's
    confidence-replacement loop never changes WHICH objects are matched, only which detection is credited)."""
    num_images = y_true[0].shape[0]
    n_true = np.zeros(num_classes, np.int64)
    n_pred = np.zeros(num_classes, np.int64)
    n_tp = np.zeros(num_classes, np.int64)
    boxes_all, confs_all, probs_all = (_to_numpy(t) for t in y_pred)
    for i in range(num_images):
        gt_labels, gt_boxes = _ground_truth_of_image(y_true, i)
        n_true += np.bincount(gt_labels, minlength=num_classes)[:num_classes]
        det_boxes, _, det_labels = nms_fn(boxes_all[i:i + 1], confs_all[i:i + 1] * probs_all[i:i + 1])
        if det_labels is None or len(det_labels) == 0:
            continue
        det_boxes, det_labels = _to_numpy(det_boxes), _to_numpy(det_labels).astype(np.int64)
        n_pred += np.bincount(det_labels, minlength=num_classes)[:num_classes]
        if gt_labels.size == 0:
            continue                      # (the reference indexes an empty array here and raises)
        iou = calc_iou(det_boxes, gt_boxes)
        best = np.argmax(iou, axis=-1)
        hit = (iou[np.arange(len(best)), best] > iou_thresh) & (gt_labels[best] == det_labels)
        found = np.unique(best[hit])
        n_tp += np.bincount(gt_labels[found], minlength=num_classes)[:num_classes]
    if calc_now:
        # avoid divided by 0
        return n_tp.sum() / (n_true.sum() + 1e-6), n_tp.sum() / (n_pred.sum() + 1e-6)
    as_dict = lambda a: {c: int(a[c]) for c in range(num_classes)}
    return as_dict(n_tp), as_dict(n_true), as_dict(n_pred)


def evaluate_on_cpu(y_pred, y_true, num_classes, calc_now=True, max_boxes=50, score_thresh=0.5, iou_thresh=0.5):
    '''
    Given y_pred (boxes [N,B,4], confs [N,B,1], probs [N,B,C]) and y_true (the three process_box tensors) of a
    batch, get the recall and precision of the batch (calc_now) or the per-class
    (true_positive_dict, true_labels_dict, pred_labels_dict).  NMS follows cpu_nms (utils/nms_utils.py:94-123).
    '''
    nms = lambda b, s: cpu_nms(b, s, num_classes, max_boxes=max_boxes, score_thresh=score_thresh,
                               iou_thresh=iou_thresh)
    return _evaluate(y_pred, y_true, num_classes, nms, iou_thresh, calc_now)


def evaluate_on_gpu(sess, gpu_nms_op, pred_boxes_flag, pred_scores_flag, y_pred, y_true, num_classes, iou_thresh=0.5,
                    calc_now=True):
    '''
    Same as evaluate_on_cpu with the NMS given by `gpu_nms_op` (a callable (boxes, scores) -> (boxes, scores,
    labels), e.g. functools.partial(gpu_nms, num_classes=C, max_boxes=.., score_thresh=.., nms_thresh=..)).
    `sess`, `pred_boxes_flag`, `pred_scores_flag` exist for signature compatibility and are ignored.
    '''
    return _evaluate(y_pred, y_true, num_classes, gpu_nms_op, iou_thresh, calc_now)


def _rows(image_id, boxes, scores, labels):
    boxes, scores, labels = _to_numpy(boxes), _to_numpy(scores), _to_numpy(labels)
    return [[image_id, boxes[k][0], boxes[k][1], boxes[k][2], boxes[k][3], scores[k], labels[k]]
            for k in range(len(labels))]


def get_preds_gpu(sess, gpu_nms_op, pred_boxes_flag, pred_scores_flag, image_ids, y_pred):
    '''
    Given the y_pred of ONE input image, get the predicted bbox and label info.
    return:
        pred_content: 2d list, rows [image_id, x_min, y_min, x_max, y_max, score, label].
    '''
    boxes, confs, probs = y_pred[0][0:1], y_pred[1][0:1], y_pred[2][0:1]
    det = gpu_nms_op(boxes, confs * probs)
    return _rows(image_ids[0], *det)


def get_preds_batch(image_ids, detections):
    """Rows of get_preds_gpu for a whole batch: `detections` is the list yolov3.detect returns (one
    (boxes, scores, labels) tuple of device tensors per image)."""
    out = []
    for image_id, det in zip(image_ids, detections):
        out += _rows(image_id, *det)
    return out


gt_dict = {}  # key: img_id, value: gt object list (module-level cache, like the reference)


def parse_gt_rec(gt_filename, target_img_size, letterbox_resize=True):
    '''
    parse and re-organize the gt info: boxes are mapped to the network input frame
    (letterbox: scale by min ratio + integer pad; else independent x/y scale).
    return:
        gt_dict: dict. Each key is a img_id, the value is the gt bboxes in the corresponding img.
    '''
    global gt_dict
    if gt_dict:
        return gt_dict
    new_width, new_height = target_img_size
    with open(gt_filename, 'r') as f:
        for line in f:
            img_id, _, boxes, labels, ori_width, ori_height = parse_line(line)
            if letterbox_resize:
                ratio = min(new_width / ori_width, new_height / ori_height)
                sx = sy = ratio
                dw = int((new_width - int(ratio * ori_width)) / 2)
                dh = int((new_height - int(ratio * ori_height)) / 2)
            else:
                sx, sy, dw, dh = None, None, 0, 0
            objects = []
            for (x_min, y_min, x_max, y_max), label in zip(boxes, labels):
                if letterbox_resize:
                    objects.append([x_min * sx + dw, y_min * sy + dh, x_max * sx + dw, y_max * sy + dh, label])
                else:
                    objects.append([x_min * new_width / ori_width, y_min * new_height / ori_height,
                                    x_max * new_width / ori_width, y_max * new_height / ori_height, label])
            gt_dict[img_id] = objects
    return gt_dict


def voc_ap(rec, prec, use_07_metric=False):
    """VOC AP from cumulative recall / precision arrays: the 11-point VOC07 metric, or (default) the area under
    the monotone precision envelope."""
    rec, prec = np.asarray(rec), np.asarray(prec)
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            above = rec >= t
            ap = ap + (np.max(prec[above]) if above.any() else 0) / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]          # precision envelope (running max from the right)
    step = np.where(mrec[1:] != mrec[:-1])[0]                # where recall changes value
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def voc_eval(gt_dict, val_preds, classidx, iou_thres=0.5, use_07_metric=False):
    '''
    PASCAL VOC evaluation of one class.
    gt_dict: parse_gt_rec's dict; val_preds: rows of get_preds_gpu over the whole set.
    returns (npos, nd, recall, precision, ap); (1e-6, 1e-6, 0, 0, 0) when the class has no detection.
    '''
    gt_boxes, gt_used, npos = {}, {}, 0
    for img_id, objs in gt_dict.items():
        mine = np.array([o[:4] for o in objs if o[-1] == classidx], dtype=np.float64).reshape(-1, 4)
        gt_boxes[img_id] = mine
        gt_used[img_id] = np.zeros(len(mine), bool)
        npos += len(mine)

    pred = [p for p in val_preds if p[-1] == classidx]
    if not pred:
        print('no box, ignore')
        return 1e-6, 1e-6, 0, 0, 0
    order = np.argsort(-np.array([p[-2] for p in pred]))      # by descending confidence
    nd = len(pred)
    tp = np.zeros(nd)
    for rank, k in enumerate(order):
        img_id = pred[k][0]
        bb = np.array(pred[k][1:5], dtype=np.float64)
        g = gt_boxes[img_id]
        if g.size == 0:
            continue
        # pixel-inclusive (+1) intersection over union, as in the VOC devkit
        iw = np.maximum(np.minimum(g[:, 2], bb[2]) - np.maximum(g[:, 0], bb[0]) + 1., 0.)
        ih = np.maximum(np.minimum(g[:, 3], bb[3]) - np.maximum(g[:, 1], bb[1]) + 1., 0.)
        inters = iw * ih
        uni = ((bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) + (g[:, 2] - g[:, 0] + 1.) * (g[:, 3] - g[:, 1] + 1.)
               - inters)
        overlaps = inters / uni
        j = int(np.argmax(overlaps))
        if overlaps[j] > iou_thres and not gt_used[img_id][j]:
            gt_used[img_id][j] = True          # first (most confident) detection of this object
            tp[rank] = 1.
    fp = np.cumsum(1. - tp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    # avoid divide by zero in case the first detection matches a difficult ground truth
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    ap = voc_ap(rec, prec, use_07_metric)
    return npos, nd, tp[-1] / float(npos), tp[-1] / float(nd), ap
