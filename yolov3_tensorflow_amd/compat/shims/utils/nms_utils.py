# coding: utf-8
"""utils.nms_utils of the reference (ref: utils/nms_utils.py:8-123): gpu_nms also accepts graph tensors."""
import numpy as _np

from yolov3_tensorflow_amd.utils import nms_utils as _nms
from yolov3_tensorflow_amd.utils.nms_utils import py_nms, cpu_nms, gpu_nms_batched      # noqa: F401
from yolov3_tensorflow_amd.compat import lazy as _lazy


def gpu_nms(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, nms_thresh=0.5):
    if not (_lazy.is_node(boxes) or _lazy.is_node(scores)):
        return _nms.gpu_nms(boxes, scores, num_classes, max_boxes, score_thresh, nms_thresh)
    empties = [_np.zeros((0, 4), _np.float32), _np.zeros((0,), _np.float32), _np.zeros((0,), _np.int32)]
    return _lazy.multi(lambda b, s: _nms.gpu_nms(b, s, num_classes, max_boxes, score_thresh, nms_thresh),
                       (boxes, scores), 3, 'gpu_nms', empties)
