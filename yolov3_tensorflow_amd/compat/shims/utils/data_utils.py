# coding: utf-8
"""utils.data_utils of the reference (ref: utils/data_utils.py:17-115) = this package's module under that name."""
from yolov3_tensorflow_amd.utils.data_utils import *          # noqa: F401,F403
