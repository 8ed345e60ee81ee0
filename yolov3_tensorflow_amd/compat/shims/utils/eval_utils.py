# coding: utf-8
"""utils.eval_utils of the reference (ref: utils/eval_utils.py:170-260) = this package's module under that name; the two
helpers that take a Session and an NMS *op* also accept the graph-tensor form the TF-1 scripts build
(`gpu_nms_op = gpu_nms(pred_boxes_flag, pred_scores_flag, ...)`, evaluated with `sess.run(..., feed_dict=...)`)."""
from yolov3_tensorflow_amd.utils.eval_utils import *          # noqa: F401,F403
from yolov3_tensorflow_amd.utils import eval_utils as _eu
from yolov3_tensorflow_amd.compat import lazy as _lazy


def _as_callable(sess, gpu_nms_op, pred_boxes_flag, pred_scores_flag):
    if not _lazy.is_node(gpu_nms_op):
        return gpu_nms_op
    return lambda boxes, scores: sess.run(list(gpu_nms_op), feed_dict={pred_boxes_flag: _eu._to_numpy(boxes),
                                                                       pred_scores_flag: _eu._to_numpy(scores)})


def get_preds_gpu(sess, gpu_nms_op, pred_boxes_flag, pred_scores_flag, image_ids, y_pred):
    return _eu.get_preds_gpu(sess, _as_callable(sess, gpu_nms_op, pred_boxes_flag, pred_scores_flag), pred_boxes_flag,
                             pred_scores_flag, image_ids, y_pred)


def evaluate_on_gpu(sess, gpu_nms_op, pred_boxes_flag, pred_scores_flag, y_pred, y_true, num_classes, iou_thresh=0.5,
                    calc_now=True):
    return _eu.evaluate_on_gpu(sess, _as_callable(sess, gpu_nms_op, pred_boxes_flag, pred_scores_flag), pred_boxes_flag,
                               pred_scores_flag, y_pred, y_true, num_classes, iou_thresh, calc_now)
