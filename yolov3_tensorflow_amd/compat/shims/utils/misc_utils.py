# coding: utf-8
"""utils.misc_utils of the reference (ref: utils/misc_utils.py) = this package's module under that name."""
from yolov3_tensorflow_amd.utils.misc_utils import *          # noqa: F401,F403
from yolov3_tensorflow_amd.utils.misc_utils import (AverageMeter, parse_anchors, read_class_names, load_weights,   # noqa: F401
                                                    save_weights, config_learning_rate, config_optimizer, Saver,
                                                    run_ops, get_variables_to_restore)
from yolov3_tensorflow_amd.utils import misc_utils as _native
from yolov3_tensorflow_amd.compat import lazy as _lazy


def config_learning_rate(args, global_step):
    """ref: utils/misc_utils.py:129-148.  With a graph tensor for `global_step` (train.py:94-98) the schedule becomes a
    graph tensor too: the native schedule evaluated at the step's value in every Session.run."""
    if not _lazy.is_node(global_step):
        return _native.config_learning_rate(args, global_step)
    if args.lr_type not in ('exponential', 'cosine_decay', 'cosine_decay_restart', 'fixed', 'piecewise'):
        raise ValueError('Unsupported learning rate type!')          # (at graph-building time, like the reference)
    return _lazy.Node(lambda g: _native.config_learning_rate(args, float(g)), (global_step,), name='learning_rate',
                      host=True, empty=0.0)


def config_optimizer(optimizer_name, learning_rate, decay=0.9, momentum=0.9):
    """ref: utils/misc_utils.py:151-161: the tf.train optimizer of the `tensorflow` shim when the learning rate is a
    graph tensor, the native optimizer object otherwise."""
    if not _lazy.is_node(learning_rate):
        return _native.config_optimizer(optimizer_name, learning_rate, decay, momentum)
    import tensorflow as tf
    if optimizer_name == 'momentum':
        return tf.train.MomentumOptimizer(learning_rate, momentum=momentum)
    elif optimizer_name == 'rmsprop':
        return tf.train.RMSPropOptimizer(learning_rate, decay=decay, momentum=momentum)
    elif optimizer_name == 'adam':
        return tf.train.AdamOptimizer(learning_rate)
    elif optimizer_name == 'sgd':
        return tf.train.GradientDescentOptimizer(learning_rate)
    else:
        raise ValueError('Unsupported optimizer type!')
