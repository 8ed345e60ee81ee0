# coding: utf-8
"""utils.data_aug of the reference under its own module path, the part the inference scripts import (ref:
utils/data_aug.py:274-320): letterbox / plain resize with the boxes following.  The training-time augmentations (mix-up,
colour jitter, expansion, constrained crop, flips) live in yolov3_tensorflow_amd.utils.data_aug and are reached through
the shimmed utils.data_utils.get_batch_data, which is what the reference's train.py calls."""
from yolov3_tensorflow_amd.utils.data_utils import letterbox_resize, resize_with_bbox      # noqa: F401
