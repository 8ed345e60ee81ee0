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
string = _np.object_

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


# ---------------------------------------------------------------------------------------------------------------
# tf.py_func / tf.data: the input pipeline of eval.py (ref: eval.py:75-92) as Python iterators behind graph tensors
# ---------------------------------------------------------------------------------------------------------------
class _Errors(object):
    class OutOfRangeError(Exception):
        """Raised by Session.run when a one-shot iterator is exhausted (tf.errors.OutOfRangeError)."""


errors = _Errors()


def py_func(func, inp, Tout, stateful=True, name=None):
    """tf.py_func: `func` runs on the host with the evaluated inputs (numpy arrays / Python values); one graph tensor
    per entry of Tout, cast to that dtype."""
    single = not isinstance(Tout, (list, tuple))
    touts = [Tout] if single else list(Tout)

    def call(*args):
        res = func(*args)
        res = [res] if single else list(res)
        if len(res) != len(touts):
            raise ValueError("py_func: the function returned %d values, Tout names %d" % (len(res), len(touts)))
        return tuple(_np.asarray(r, dtype=t) for r, t in zip(res, touts))

    host = all(getattr(a, 'host', True) for a in inp)          # constants and host tensors only -> runs in a dry run too
    outs = _lazy.multi(call, tuple(inp), len(touts), name or 'PyFunc', [_np.zeros((0,), t) for t in touts], host=host)
    return outs[0] if single else list(outs)


class _Dataset(object):
    """The slice of tf.data the reference uses: TextLineDataset -> shuffle / batch / map / prefetch -> one-shot
    iterator.  `source()` returns a fresh Python iterator of elements; `graph` maps an element tensor to output tensors."""

    def __init__(self, source, graph=None):
        self._source, self._graph = source, graph

    def batch(self, batch_size, drop_remainder=False):
        src = self._source

        def batched():
            chunk = []
            for item in src():
                chunk.append(item)
                if len(chunk) == batch_size:
                    yield _np.asarray(chunk, dtype=object)
                    chunk = []
            if chunk and not drop_remainder:
                yield _np.asarray(chunk, dtype=object)
        return _Dataset(batched, self._graph)

    def shuffle(self, buffer_size, seed=None, reshuffle_each_iteration=None):
        src = self._source

        def shuffled():
            items = list(src())
            _np.random.RandomState(seed).shuffle(items)
            return iter(items)
        return _Dataset(shuffled, self._graph)

    def map(self, map_func, num_parallel_calls=None):
        prev = self._graph
        return _Dataset(self._source, (lambda x: map_func(prev(x))) if prev else map_func)

    def prefetch(self, buffer_size):
        return self

    def repeat(self, count=None):
        src = self._source

        def repeated():
            n = 0
            while count is None or n < count:
                for item in src():
                    yield item
                n += 1
        return _Dataset(repeated, self._graph)

    def make_one_shot_iterator(self):
        return _Iterator(self)


class _Iterator(object):
    def __init__(self, dataset):
        self._dataset, self._it = dataset, None

    def get_next(self, name=None):
        def pull():
            if self._it is None:
                self._it = iter(self._dataset._source())
            try:
                return next(self._it)
            except StopIteration:
                raise errors.OutOfRangeError('End of sequence')
        element = _lazy.Node(pull, (), name='IteratorGetNext', empty=_np.zeros((0,), object), host=True)
        return self._dataset._graph(element) if self._dataset._graph else element


class _Data(object):
    Dataset = _Dataset

    @staticmethod
    def TextLineDataset(filenames):
        names = [filenames] if isinstance(filenames, str) else list(filenames)

        def lines():
            for name in names:
                with open(name, 'rb') as f:
                    for line in f:
                        yield line.rstrip(b'\r\n')
        return _Dataset(lines)


data = _Data()
