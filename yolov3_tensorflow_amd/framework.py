"""Minimal variable/scope plumbing the reference's call sites rely on (tf.variable_scope,
tf.global_variables(scope=...), slim's Conv/Conv_1/... auto-naming) plus device-context handling.

torch is used only as the owner of device memory and streams: every tensor handed to the C ABI is a
contiguous fp32 torch tensor on a HIP device, passed as a raw pointer.
"""
import contextlib
import itertools
import ctypes
import threading

import numpy as np
import torch

from . import _lib

_state = threading.local()


def _scope_stack():
    if not hasattr(_state, "stack"):
        _state.stack = []      # list of (name, counters dict)
    return _state.stack


_VERSION_COUNTER = itertools.count(1)


class Variable(object):
    """A named, device-resident fp32 parameter (the analogue of a tf.Variable).

    `version` changes on every assignment and is unique across ALL variables of the process (one global counter), so
    that (op_name, version) identifies a value even after reset_default_graph() re-creates a variable of the same
    name — the packed-weight caches are keyed by it."""

    def __init__(self, name, shape, trainable=True):
        self.name = name                      # e.g. 'yolov3/darknet53_body/Conv_1/weights:0'
        self.shape = _Shape(shape)
        self.trainable = trainable
        self.tensor = None                    # torch tensor on the device, created by .assign()
        self.version = 0

    @property
    def op_name(self):
        return self.name[:-2] if self.name.endswith(":0") else self.name

    def assign(self, value, validate_shape=True):
        arr = np.ascontiguousarray(np.asarray(value, dtype=np.float32)) if not torch.is_tensor(value) else value
        if validate_shape and tuple(arr.shape) != tuple(self.shape.as_list()):
            # mirrors tf.assign(validate_shape=True) -> ValueError on mismatch (utils/misc_utils.py:99)
            raise ValueError("Shapes %s and %s are incompatible for %s" %
                             (tuple(self.shape.as_list()), tuple(arr.shape), self.name))
        dev = default_device()
        if torch.is_tensor(arr):
            t = arr.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
        else:
            t = torch.from_numpy(arr).to(dev)
        self.tensor = t
        self.touch()
        return self

    def touch(self):
        """Mark the value as changed (after an in-place update of .tensor)."""
        self.version = next(_VERSION_COUNTER)
        _bump_global_version()

    def numpy(self):
        return self.tensor.detach().cpu().numpy()

    def __repr__(self):
        return "<y3.Variable %s shape=%s>" % (self.name, self.shape.as_list())


class _Shape(tuple):
    def as_list(self):
        return list(self)


_VARIABLES = {}       # op_name -> Variable, insertion-ordered (= creation order)
_GLOBAL_VERSION = [0]


def _bump_global_version():
    _GLOBAL_VERSION[0] += 1


def global_version():
    return _GLOBAL_VERSION[0]


def reset_default_graph():
    """Drop every variable (tf.reset_default_graph analogue; used by tests)."""
    _VARIABLES.clear()
    _bump_global_version()


def global_variables(scope=None):
    """tf.global_variables(scope=...): creation-ordered variables whose name starts with `scope`."""
    if scope is None:
        return list(_VARIABLES.values())
    return [v for k, v in _VARIABLES.items() if k.startswith(scope)]


def trainable_variables(scope=None):
    return [v for v in global_variables(scope) if v.trainable]


@contextlib.contextmanager
def variable_scope(name):
    """tf.variable_scope(name): nests names with '/', and restarts slim's layer auto-numbering."""
    stack = _scope_stack()
    stack.append((name, {}))
    try:
        yield
    finally:
        stack.pop()


@contextlib.contextmanager
def variable_scope_absolute(name):
    """Run the block under exactly the scope `name` ('' = the root), whatever scopes are open around it: deferred
    graph pieces (compat layer) are evaluated long after their `with variable_scope(...)` block was left.
    Like a fresh `with variable_scope(name):` at the root, the block starts with NEW auto-naming counters (the first conv
    is 'Conv' again): variables are created-or-reused by name (get_variable), so re-entering a scope re-finds the same
    variables; the surrounding scopes and their counters come back untouched afterwards."""
    stack = _scope_stack()
    saved = list(stack)
    stack[:] = [(name, {})] if name else []
    try:
        yield
    finally:
        stack[:] = saved


def current_scope_name():
    return "/".join(n for n, _ in _scope_stack())


def unique_layer_name(base):
    """slim auto-naming inside the innermost scope: Conv, Conv_1, Conv_2, ..."""
    stack = _scope_stack()
    if not stack:
        stack.append(("", {}))
    counters = stack[-1][1]
    k = counters.get(base, 0)
    counters[base] = k + 1
    return base if k == 0 else "%s_%d" % (base, k)


def get_variable(full_name, shape, initializer, trainable=True):
    """Create-or-reuse (AUTO_REUSE semantics: eager re-invocation of forward() reuses the variables)."""
    v = _VARIABLES.get(full_name)
    if v is not None:
        if tuple(v.shape) != tuple(shape):
            raise ValueError("Trying to share variable %s, but specified shape %s and found shape %s." %
                             (full_name, tuple(shape), tuple(v.shape)))
        return v
    v = Variable(full_name + ":0", shape, trainable)
    v.assign(initializer(shape))
    _VARIABLES[full_name] = v
    return v


# ---- initialisers (slim defaults: xavier-uniform kernels, BN gamma=1 beta=0 mean=0 var=1) ----------
_init_rng = np.random.RandomState(0)


def set_init_seed(seed):
    global _init_rng
    _init_rng = np.random.RandomState(seed)


def xavier_uniform(shape):
    kh, kw, cin, cout = shape
    limit = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
    return _init_rng.uniform(-limit, limit, size=shape).astype(np.float32)


def zeros(shape):
    return np.zeros(shape, np.float32)


def ones(shape):
    return np.ones(shape, np.float32)


# ---- device / stream context ----------------------------------------------------------------------
_default_device = [None]
_ctx_cache = {}


def set_default_device(dev):
    _default_device[0] = torch.device(dev)


def default_device():
    if _default_device[0] is None:
        if not torch.cuda.is_available():
            raise _lib.Y3Error("no HIP device is visible: the yolo355 hot path runs only on an MI355X "
                               "(there is no CPU fallback)")
        _default_device[0] = torch.device("cuda", torch.cuda.current_device())
    return _default_device[0]


def context(device=None):
    """y3_ctx for (device, torch's current stream on it); cached."""
    dev = torch.device(device) if device is not None else default_device()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx).cuda_stream
    key = (idx, stream)
    ctx = _ctx_cache.get(key)
    if ctx is None:
        out = ctypes.c_void_p()
        _lib.check(_lib.lib().y3_ctx_create(idx, ctypes.c_void_p(stream), ctypes.byref(out)))
        ctx = out
        _ctx_cache[key] = ctx
    return ctx


def check_context(device=None):
    """y3_ctx_check for the current (device, stream): synchronises the stream and raises Y3Error if a kernel of an
    earlier launch reported a failure (a stream-K hand-off that timed out: include/yolo355.h).  Cheap to call wherever
    the host synchronises anyway."""
    _lib.check(_lib.lib().y3_ctx_check(context(device)))


def ptr(t):
    """Raw device pointer of a contiguous fp32/int32 tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor handed to the C ABI must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def as_device_f32(x, name="input"):
    """Accept a numpy array or torch tensor, return a contiguous fp32 tensor on the default device."""
    if torch.is_tensor(x):
        t = x
        if t.dtype != torch.float32:
            t = t.float()
        if t.device.type != "cuda":
            t = t.to(default_device())
        return t.contiguous()
    arr = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    return torch.from_numpy(arr).to(default_device())
