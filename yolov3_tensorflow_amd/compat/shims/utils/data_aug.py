# coding: utf-8
"""utils.data_aug of the reference, the part the inference scripts use (ref: utils/data_aug.py:274-320): letterbox /
plain resize with the boxes following.  The training-time augmentations (mix-up, colour jitter, random crops) are out
of scope."""
from yolov3_tensorflow_amd.utils.data_utils import letterbox_resize, resize_with_bbox      # noqa: F401
