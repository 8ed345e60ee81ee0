"""yolov3_tensorflow_amd — MI355X (gfx950) native YOLOv3 hot path behind the API surface of
wizyoung/YOLOv3_TensorFlow (model.yolov3, utils.layer_utils, utils.nms_utils, utils.misc_utils).

Python here is host plumbing only (variables, scopes, device memory via torch); all arithmetic runs in
hand-written HIP kernels behind the C ABI in include/yolo355.h (csrc/libyolo355.so).
"""
from .framework import (Variable, variable_scope, variable_scope_absolute, global_variables,  # noqa: F401
                        trainable_variables, reset_default_graph, set_default_device, set_init_seed)
from .model import yolov3  # noqa: F401

__all__ = ["yolov3", "Variable", "variable_scope", "variable_scope_absolute", "global_variables", "trainable_variables",
           "reset_default_graph", "set_default_device", "set_init_seed"]
