"""yolov3_tensorflow_amd — MI355X (gfx950) native YOLOv3 hot path behind the API surface of
wizyoung/YOLOv3_TensorFlow (model.yolov3, utils.layer_utils, utils.nms_utils, utils.misc_utils).

Python here is host plumbing only (variables, scopes, device memory via torch); all arithmetic runs in
hand-written HIP kernels behind the C ABI in include/yolo355.h (csrc/libyolo355.so).
"""
import os as _os

# Two HIP streams of one GPU overlap only when they sit on different hardware queues; with the runtime's default of 4 queues (and
# RCCL initialised in the process) the side stream of the two-stream inference forward (model.inference_streams = 2) was seen
# to alias.  Ask for 8 unless the caller chose a value: it takes effect if this package is imported before the process's first
# HIP call, which is the usual order (bench.py used to be the only caller that did this).  It selects no
# kernel and changes no result; inference_streams itself stays opt-in / measured (yolov3.choose_inference_streams).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from .framework import (Variable, variable_scope, variable_scope_absolute, global_variables,  # noqa: F401
                        trainable_variables, reset_default_graph, set_default_device, set_init_seed)
from .model import yolov3  # noqa: F401

__all__ = ["yolov3", "Variable", "variable_scope", "variable_scope_absolute", "global_variables", "trainable_variables",
           "reset_default_graph", "set_default_device", "set_init_seed"]
