# coding: utf-8
"""`tensorflow` as the reference's driver scripts see it (the test_single_image.py and
convert_weight.py rows): a deferred graph over yolov3_tensorflow_amd's eager ops.  Only reachable through
`python -m yolov3_tensorflow_amd.compat.run` (or compat.install()); see yolov3_tensorflow_amd/compat/__init__.py."""
import numpy as _np

import yolov3_tensorflow_amd as _y3
from yolov3_tensorflow_amd import compat as _compat
from yolov3_tensorflow_amd.compat import lazy as _lazy

__version__ = '1.15.0-yolo355-compat'

float32, float64, int32, int64, uint8, bool = _np.float32, _np.float64, _np.int32, _np.int64, _np.uint8, _np.bool_

Tensor = _lazy.Node


def placeholder(dtype, shape=None, name=None):
    """tf.placeholder: a graph input fed through Session.run(feed_dict=...)."""
    return _lazy.Placeholder(dtype, shape, name)


def constant(value, dtype=None, shape=None, name='Const'):
    arr = _np.asarray(value, dtype=dtype)
    if shape is not None:
        arr = _np.broadcast_to(arr, shape).copy()
    return _lazy.Node(lambda: arr, (), name=name, shape=list(arr.shape), empty=arr)


def variable_scope(name_or_scope, *args, **kwargs):
    return _y3.variable_scope(name_or_scope)


def global_variables(scope=None):
    return _y3.global_variables(scope=scope)


def trainable_variables(scope=None):
    return _y3.trainable_variables(scope=scope)


def reset_default_graph():
    _y3.reset_default_graph()


class _NoOp(object):
    def run(self):
        pass


def global_variables_initializer():
    """Variables are initialised where they are created (slim's initialisers, model.py:35-49); nothing is left to do."""
    return _NoOp()


local_variables_initializer = global_variables_initializer


class Session(object):
    """tf.Session: `run` evaluates graph tensors with this package's device ops and returns numpy arrays."""

    def __init__(self, target='', graph=None, config=None):
        self._closed = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        self._closed = True

    def run(self, fetches, feed_dict=None, options=None, run_metadata=None):
        if self._closed:
            raise RuntimeError('Attempted to use a closed Session.')
        single = not isinstance(fetches, (list, tuple))
        res = _lazy.evaluate([fetches] if single else list(fetches), feed_dict, dry=_compat.dry_run())
        return res[0] if single else res


InteractiveSession = Session


class ConfigProto(object):
    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)


class _Saver(object):
    """tf.train.Saver over the native checkpoint (utils.misc_utils.Saver: one .npz keyed by the TF variable names).
    `restore` also accepts a darknet .weights file: TF's own checkpoint format cannot be read without TensorFlow."""

    def __init__(self, var_list=None, max_to_keep=5, **kwargs):
        self._var_list = var_list

    def _vars(self):
        return list(self._var_list) if self._var_list is not None else _y3.global_variables()

    def save(self, sess, save_path, global_step=None, **kwargs):
        from yolov3_tensorflow_amd.utils import misc_utils
        if _compat.dry_run():
            return save_path
        path = save_path if global_step is None else '%s-%d' % (save_path, int(global_step))
        misc_utils.Saver(self._vars()).save(path)
        return path

    def restore(self, sess, save_path):
        import os
        from yolov3_tensorflow_amd.utils import misc_utils
        if _compat.dry_run():
            return
        if str(save_path).endswith('.weights'):
            misc_utils.run_ops(misc_utils.load_weights(self._vars(), save_path))
            return
        if not (os.path.exists(save_path) or os.path.exists(str(save_path) + '.npz')):
            raise IOError("The passed save_path is not a valid checkpoint: %s (native checkpoints are <path>.npz, "
                          "written by tf.train.Saver.save of this layer or utils.misc_utils.Saver)" % save_path)
        misc_utils.Saver(self._vars()).restore(save_path)


class _Train(object):
    Saver = _Saver


train = _Train()
