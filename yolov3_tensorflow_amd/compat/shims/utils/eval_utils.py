# coding: utf-8
"""utils.eval_utils of the reference (ref: utils/eval_utils.py) = this package's module under that name."""
from yolov3_tensorflow_amd.utils.eval_utils import *          # noqa: F401,F403
