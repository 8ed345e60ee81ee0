# coding: utf-8
"""utils.misc_utils of the reference (ref: utils/misc_utils.py) = this package's module under that name."""
from yolov3_tensorflow_amd.utils.misc_utils import *          # noqa: F401,F403
from yolov3_tensorflow_amd.utils.misc_utils import (AverageMeter, parse_anchors, read_class_names, load_weights,   # noqa: F401
                                                    save_weights, config_learning_rate, config_optimizer, Saver,
                                                    run_ops, get_variables_to_restore)
