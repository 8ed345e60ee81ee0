# coding: utf-8
"""The deferred graph behind the `tensorflow` shim: a node is a function of other nodes, `evaluate` walks the graph
once per Session.run with the feed values, memoising every node, and hands back numpy arrays."""
import numpy as np


class Node(object):
    """One graph tensor.  fn(*evaluated args) -> value; `index` picks one output of a multi-output op; `empty` is what
    a dry run (no device) returns for it."""

    def __init__(self, fn, args=(), name=None, index=None, empty=None, shape=None, host=False):
        self.fn, self.args, self.name, self.index, self.empty, self._shape = fn, tuple(args), name, index, empty, shape
        self.dtype = None
        self.host = host          # pure host work (input pipeline): also evaluated in a dry run
        self.late = False         # an update op (the train op): fetched AFTER every other fetch of the same run

    # the arithmetic the driver scripts apply to graph tensors (ref: test_single_image.py:55 `pred_confs * pred_probs`)
    def __mul__(self, other):
        return Node(lambda a, b: a * b, (self, other), name='mul')

    __rmul__ = __mul__

    def __add__(self, other):
        return Node(lambda a, b: a + b, (self, other), name='add')

    __radd__ = __add__

    def __sub__(self, other):
        return Node(lambda a, b: a - b, (self, other), name='sub')

    def __truediv__(self, other):
        return Node(lambda a, b: a / b, (self, other), name='div')

    def __rsub__(self, other):
        return Node(lambda a, b: b - a, (self, other), name='rsub')

    def __rtruediv__(self, other):
        return Node(lambda a, b: b / a, (self, other), name='rdiv')

    __div__, __rdiv__ = __truediv__, __rtruediv__

    def __getitem__(self, key):
        return Node(lambda a: a[key], (self,), name='getitem')

    def get_shape(self):
        return self._shape

    shape = property(get_shape)

    def set_shape(self, shape):
        self._shape = list(shape)

    def __repr__(self):
        return '<compat graph tensor %s>' % (self.name or 'op')


class Placeholder(Node):
    def __init__(self, dtype, shape=None, name=None):
        Node.__init__(self, None, (), name=name or 'Placeholder', shape=None if shape is None else list(shape))
        self.dtype = dtype


def find(x, pred, _seen=None):
    """First node reachable from `x` (through node arguments and lists) that satisfies `pred`, or None."""
    _seen = set() if _seen is None else _seen
    if isinstance(x, (list, tuple)):
        for v in x:
            r = find(v, pred, _seen)
            if r is not None:
                return r
        return None
    if not isinstance(x, Node) or id(x) in _seen:
        return None
    _seen.add(id(x))
    if pred(x):
        return x
    return find(x.args, pred, _seen)


def is_node(x):
    if isinstance(x, Node):
        return True
    if isinstance(x, (list, tuple)):
        return any(is_node(v) for v in x)
    return False


def multi(fn, args, nout, name, empties=None, host=False):
    """An op with `nout` outputs: one hidden node computing the tuple, `nout` visible nodes indexing it."""
    whole = Node(fn, args, name=name, host=host)
    return tuple(Node(lambda t, i=i: t[i], (whole,), name='%s:%d' % (name, i), index=i,
                      empty=None if empties is None else empties[i], host=host) for i in range(nout))


def _to_numpy(v):
    try:
        import torch
        if isinstance(v, torch.Tensor):
            return v.detach().cpu().numpy()
    except ImportError:
        pass
    return v


def evaluate(fetches, feed_dict=None, dry=False):
    """Session.run: `fetches` is a node, an op with .run(), or a (nested) list of those; returns numpy arrays in the
    same structure (None for ops)."""
    feed = {}
    for k, v in (feed_dict or {}).items():
        if not isinstance(k, Placeholder):
            raise TypeError("feed_dict keys must be placeholders, got %r" % (k,))
        feed[id(k)] = v
    memo = {}

    def ev(x):
        if isinstance(x, (list, tuple)):
            return type(x)(ev(v) for v in x) if isinstance(x, tuple) else [ev(v) for v in x]
        if not isinstance(x, Node):
            return x
        key = id(x)
        if key in memo:
            return memo[key]
        if isinstance(x, Placeholder):
            if key not in feed:
                raise ValueError("You must feed a value for placeholder tensor '%s'" % x.name)
            val = np.asarray(feed[key])
            if x._shape is not None and len(x._shape) == val.ndim:
                for want, got in zip(x._shape, val.shape):
                    if want is not None and want != got:
                        raise ValueError("Cannot feed value of shape %s for Tensor '%s', which has shape %s"
                                         % (val.shape, x.name, tuple(x._shape)))
        else:
            val = x.fn(*[ev(a) for a in x.args])
        memo[key] = val
        return val

    done = {}

    def out(x, late):
        if isinstance(x, (list, tuple)):
            return [out(v, late) for v in x]
        if isinstance(x, Node):
            if x.late != late:
                return done.get(id(x))
            if id(x) not in done:
                done[id(x)] = x.empty if (dry and not x.host) else _to_numpy(ev(x))
            return done[id(x)]
        if hasattr(x, 'op_name') and hasattr(x, 'tensor'):       # a model variable: its current value
            if id(x) not in done:
                done[id(x)] = None if dry else _to_numpy(x.tensor)
            return done[id(x)]
        if hasattr(x, 'run'):          # an assign op (utils.misc_utils.AssignOp), an iterator initializer, a group of them
            if not late and (not dry or getattr(x, 'host', False)):
                x.run()
            return None
        raise TypeError("Fetch argument %r cannot be interpreted as a graph tensor or an op" % (x,))

    # reads first, updates second: what the other fetches of a run see (feature maps, loss, global_step) is the state
    # BEFORE the train op of the same run changed it - and the train op may reuse the memory they live in
    out(fetches, False)
    return out(fetches, True)
