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
        self.graph = graph           # (only handed to tf.summary.FileWriter, ref: train.py:126)

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
    for o, t in zip(outs, touts):
        o.dtype = t
    return outs[0] if single else list(outs)


class _Dataset(object):
    """The slice of tf.data the reference uses: TextLineDataset -> shuffle / batch -> map -> prefetch -> iterator
    (ref: train.py:34-58, eval.py:75-92).  `source()` returns a fresh Python iterator of elements; `graph` maps an element
    tensor to output tensors.  Supported pipeline shape: shuffle / batch / repeat BEFORE map (the reference's order); a
    shuffle / batch / repeat after a map raises instead of silently running in the wrong order."""

    def __init__(self, source, graph=None, n_out=None):
        self._source, self._graph, self._n_out = source, graph, n_out

    def _no_map_yet(self, what):
        if self._graph is not None:
            raise NotImplementedError("compat tf.data: %s() after map() is not supported (only the reference's "
                                      "shuffle -> batch -> map -> prefetch order is)" % what)

    def batch(self, batch_size, drop_remainder=False):
        self._no_map_yet('batch')
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
        return _Dataset(batched)

    def shuffle(self, buffer_size, seed=None, reshuffle_each_iteration=None):
        self._no_map_yet('shuffle')
        src = self._source

        def shuffled():      # (a buffer as large as the file, as train.py:35 asks: a full shuffle per iteration)
            items = list(src())
            _np.random.RandomState(seed).shuffle(items)
            return iter(items)
        return _Dataset(shuffled)

    def map(self, map_func, num_parallel_calls=None):
        """The map function becomes part of the element's graph and runs when a Session.run asks for the element.
        num_parallel_calls: the reference maps py_func(get_batch_data) over BATCHES of lines (train.py:37-43); the
        parallelism is handed to get_batch_data, which runs that many of a batch's samples at once
        (utils.data_utils.BATCH_WORKERS) - same throughput lever, no speculative evaluation of graph nodes."""
        if num_parallel_calls and int(num_parallel_calls) > 1:
            from yolov3_tensorflow_amd.utils import data_utils as _du
            _du.BATCH_WORKERS = max(int(_du.BATCH_WORKERS), int(num_parallel_calls))
        prev = self._graph
        return _Dataset(self._source, (lambda x: map_func(prev(x))) if prev else map_func)

    def prefetch(self, buffer_size):
        """Accepted; elements are produced when a run asks for them (nothing is computed ahead of the session)."""
        return self

    def repeat(self, count=None):
        self._no_map_yet('repeat')
        src = self._source

        def repeated():
            n = 0
            while count is None or n < count:
                for item in src():
                    yield item
                n += 1
        return _Dataset(repeated)

    def _probe(self):
        """The output tensors of the map graph on a dummy element (structure only: nothing is evaluated)."""
        if self._graph is None:
            return None
        out = self._graph(_lazy.Node(lambda: None, (), name='element', host=True))
        return list(out) if isinstance(out, (list, tuple)) else [out]

    @property
    def output_types(self):
        out = self._probe()
        return tuple(getattr(o, 'dtype', None) for o in out) if out else string

    @property
    def output_shapes(self):
        out = self._probe()
        return tuple(None for _ in out) if out else None

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


class _ReinitIterator(object):
    """tf.data.Iterator.from_structure(types, shapes): ONE get_next() whose source is whichever dataset the last
    initializer op selected (ref: train.py:55-60: the train / validation switch)."""

    def __init__(self, output_types, output_shapes=None):
        self._types = output_types if isinstance(output_types, (list, tuple)) else (output_types,)
        self._dataset, self._it = None, None

    def make_initializer(self, dataset, name=None):
        it = self

        class _Init(object):
            host = True              # (input pipeline: runs in a dry run too)

            def run(self_inner):
                it._dataset, it._it = dataset, iter(dataset._source())
        return _Init()

    def get_next(self, name=None):
        n = len(self._types)

        def pull_and_map():
            if self._dataset is None:
                raise RuntimeError("GetNext() failed because the iterator has not been initialized")
            try:
                item = next(self._it)
            except StopIteration:
                raise errors.OutOfRangeError('End of sequence')
            if self._dataset._graph is None:
                return (item,)
            out = self._dataset._graph(_lazy.Node(lambda: item, (), name='element', host=True))
            out = list(out) if isinstance(out, (list, tuple)) else [out]
            return tuple(_lazy.evaluate(out, None, dry=False))       # (host pipeline: evaluated in a dry run too)

        empties = [_np.zeros((0,), _np.int64)] + [_np.zeros((0, 0, 0, 3), _np.float32)] * max(n - 1, 0)
        outs = _lazy.multi(pull_and_map, (), n, 'IteratorGetNext', empties[:n], host=True)
        return outs if n > 1 else outs[0]


class _IteratorNS(object):
    from_structure = staticmethod(lambda output_types, output_shapes=None, **kw: _ReinitIterator(output_types, output_shapes))


class _Data(object):
    Dataset = _Dataset
    Iterator = _IteratorNS

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


# ---------------------------------------------------------------------------------------------------------------
# the training graph of train.py:72-124: global step, tf.cond, learning-rate schedule, optimizer ops, collections,
# control dependencies, summaries.  One Session.run of the train op = one step of yolov3_tensorflow_amd.training.Trainer
# on the forward / loss state of the same run.
# ---------------------------------------------------------------------------------------------------------------
class GraphKeys(object):
    GLOBAL_VARIABLES, LOCAL_VARIABLES, TRAINABLE_VARIABLES = 'variables', 'local_variables', 'trainable_variables'
    UPDATE_OPS, REGULARIZATION_LOSSES, SUMMARIES = 'update_ops', 'regularization_losses', 'summaries'


class _ScalarVariable(_lazy.Node):
    """tf.Variable(scalar, trainable=False, ...): train.py:92's float global_step (lives on the host)."""

    def __init__(self, initial_value, trainable=True, collections=None, name=None, dtype=None):
        _lazy.Node.__init__(self, lambda: self.value, (), name=name or 'Variable', host=True)
        self.value = _np.float32(initial_value) if isinstance(initial_value, float) else initial_value
        self.empty = self.value

    def assign_add(self, delta):
        self.value = type(self.value)(self.value + delta)


def Variable(initial_value, trainable=True, collections=None, name=None, dtype=None, **kwargs):
    if _np.ndim(initial_value) != 0:
        raise NotImplementedError("compat tf.Variable: only the scalar global_step of train.py:92 is supported; model "
                                  "variables are created by model.yolov3.forward")
    return _ScalarVariable(initial_value, trainable, collections, name, dtype)


def get_collection(key, scope=None):
    """UPDATE_OPS: the moving-average updates of the 72 BN layers run inside the training forward itself
    (y3_net_train_forward), so there is nothing left to depend on; the other collections the scripts never read."""
    if key == GraphKeys.TRAINABLE_VARIABLES:
        return trainable_variables(scope)
    if key == GraphKeys.GLOBAL_VARIABLES:
        return global_variables(scope)
    return []


class control_dependencies(object):
    def __init__(self, control_inputs):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def less(x, y, name=None):
    return _lazy.Node(lambda a, b: a < b, (x, y), name='Less', host=True)


def cond(pred, true_fn=None, false_fn=None, name=None, **kwargs):
    """tf.cond: both branches are BUILT (like TF), only the taken one is evaluated."""
    t, f = true_fn(), false_fn()

    def run(p):
        return _lazy.evaluate([t if bool(p) else f], None, dry=False)[0]
    return _lazy.Node(run, (pred,), name='cond', host=True, empty=_np.float32(0.0))


def clip_by_norm(t, clip_norm, axes=None, name=None):
    """tf.clip_by_norm on a gradient of the compat optimizer: records the norm on it - the clip runs inside
    y3_clip_update_multi with that value (train.py:113-114)."""
    if isinstance(t, _GradNode):
        return _GradNode(t.var, t.optimizer, float(clip_norm))
    return _lazy.Node(lambda a: a * min(1.0, float(clip_norm) / max(float(_np.sqrt((a * a).sum())), 1e-30)), (t,), name='clip_by_norm')


class _GradNode(_lazy.Node):
    """d loss / d var as a graph tensor: only its identity matters (the gradient itself lives in the Trainer's flat
    buffer); evaluating it returns the variable's gradient view after the step."""

    def __init__(self, var, optimizer, clip=None):
        _lazy.Node.__init__(self, lambda: None, (), name='gradients/' + var.op_name, empty=_np.zeros((0,), _np.float32))
        self.var, self.optimizer, self.clip = var, optimizer, clip


class _Optimizer(object):
    """tf.train.{GradientDescent,Momentum,Adam,RMSProp}Optimizer as train.py uses them (compute_gradients ->
    clip_by_norm -> apply_gradients, or minimize): the ops are carried out by yolov3_tensorflow_amd.training.Trainer."""

    def __init__(self, kind, learning_rate, **hyper):
        self.kind, self.learning_rate, self.hyper = kind, learning_rate, hyper
        self._loss, self._var_list, self._trainer = None, None, None

    def compute_gradients(self, loss, var_list=None, **kwargs):
        self._loss = loss
        self._var_list = list(var_list) if var_list is not None else trainable_variables()
        return [(_GradNode(v, self), v) for v in self._var_list if getattr(v, 'trainable', True)]

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gvs = [gv for gv in grads_and_vars if gv[0] is not None]
        clips = set(g.clip for g, _ in gvs if isinstance(g, _GradNode))
        if len(clips) > 1:
            raise NotImplementedError("compat optimizer: one clip norm for all gradients (train.py:113 uses 100.)")
        clip = clips.pop() if clips else None
        update_vars = [v for _, v in gvs]
        opt = self

        def step(loss_value, lr_value):
            from yolov3_tensorflow_amd import training
            model_node = _lazy.find(opt._loss, lambda n: getattr(n, 'model', None) is not None)
            if model_node is None:
                raise RuntimeError("compat optimizer: the loss does not come from yolov3.compute_loss")
            model = model_node.model
            if opt._trainer is None:
                o = training.Optimizer(opt.kind, float(lr_value), **opt.hyper)
                opt._trainer = training.Trainer(model, o, update_vars=update_vars,
                                                clip_norm=clip if clip is not None else float('inf'),
                                                global_step=float(global_step.value) if global_step is not None else 0.0)
            tr = opt._trainer
            tr.opt.learning_rate = float(lr_value)
            tr.backward()
            tr.apply_gradients()
            if global_step is not None:
                global_step.assign_add(1)
            return None
        op = _lazy.Node(step, (self._loss, self.learning_rate), name='train_op', empty=None)
        op.late = True
        return op

    def minimize(self, loss, global_step=None, var_list=None, **kwargs):
        return self.apply_gradients(self.compute_gradients(loss, var_list=var_list), global_step=global_step)


_Train.GradientDescentOptimizer = staticmethod(lambda learning_rate, **kw: _Optimizer('sgd', learning_rate))
_Train.MomentumOptimizer = staticmethod(lambda learning_rate, momentum=0.9, **kw: _Optimizer('momentum', learning_rate, momentum=momentum))
_Train.AdamOptimizer = staticmethod(lambda learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **kw: _Optimizer(
    'adam', learning_rate, beta1=beta1, beta2=beta2, epsilon=epsilon))
_Train.RMSPropOptimizer = staticmethod(lambda learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, **kw: _Optimizer(
    'rmsprop', learning_rate, decay=decay, momentum=momentum, epsilon=epsilon))


class _Losses(object):
    @staticmethod
    def get_regularization_loss(scope=None, name='total_regularization_loss'):
        """sum over the conv kernels of weight_decay * ||w||^2 / 2 (slim's l2_regularizer, model.py:49).  Only a
        DIAGNOSTIC here (train.py:78,90-91 put it in summaries): the train op adds weight_decay * w to the gradients
        inside y3_clip_update_multi, which is the gradient of exactly this term."""
        def run():
            import torch
            total, wd = 0.0, 5e-4
            for v in global_variables():
                if v.op_name.endswith('/weights'):
                    total += float((v.tensor.double() ** 2).sum().item())
            return _np.float32(0.5 * _REG['weight_decay'] * total)
        return _lazy.Node(run, (), name='total_regularization_loss', empty=_np.float32(0.0))


_REG = {'weight_decay': 5e-4}      # model.yolov3(..., weight_decay) of the compat model shim records its value here
losses = _Losses()


class _Summary(object):
    """tf.summary: scalars are recorded as (tag, tensor) and the merged op evaluates to the empty serialized summary -
    nothing is fetched for it (no TensorBoard writer without TensorFlow); FileWriter keeps what add_summary is handed."""

    def __init__(self):
        self.scalars = []

    def scalar(self, name, tensor, **kwargs):
        self.scalars.append((name, tensor))
        return _lazy.Node(lambda: b'', (), name='summary/' + name, empty=b'', host=True)

    def merge_all(self, *args, **kwargs):
        return _lazy.Node(lambda: b'', (), name='Merge/MergeSummary', empty=b'', host=True)

    class FileWriter(object):
        def __init__(self, logdir=None, graph=None, **kwargs):
            self.logdir, self.records = logdir, []

        def add_summary(self, summary, global_step=None):
            self.records.append((summary, global_step))

        def flush(self):
            pass

        close = flush


summary = _Summary()


class _Framework(object):
    @staticmethod
    def get_variables_to_restore(include=None, exclude=None):
        from yolov3_tensorflow_amd.utils.misc_utils import get_variables_to_restore as _g
        return _g(global_variables(), include, exclude)


class _Contrib(object):
    framework = _Framework()


contrib = _Contrib()
