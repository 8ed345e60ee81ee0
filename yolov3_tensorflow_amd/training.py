"""Training path (reference train.py:72-115): yolov3.forward(is_training=True) with batch-statistics BN,
compute_loss, backward, per-tensor clip, optimizer update, optional data-parallel gradient all-reduce.

Python here only sequences C-ABI calls over caller-owned device buffers (torch = allocator + process group);
every number is produced by a HIP kernel of csrc/y3_train.hip, y3_wgrad.hip or y3_conv.hip.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import distributed
from . import framework as fw

BN_EPS = 1e-5      # model.py:37
CLIP_NORM = 100.0  # train.py:113-114


class Optimizer(object):
    """Hyper-parameters + slot variables of one of the four update rules the reference can select
    (utils/misc_utils.py:151-161; TF1 definitions)."""

    KINDS = ('sgd', 'momentum', 'adam', 'rmsprop')

    def __init__(self, kind, learning_rate, momentum=0.9, decay=0.9, beta1=0.9, beta2=0.999, epsilon=None):
        if kind not in self.KINDS:
            raise ValueError('Unsupported optimizer type!')
        self.kind = kind
        self.learning_rate = learning_rate   # float, or a callable global_step -> float
        self.momentum = momentum
        self.decay = decay
        self.beta1, self.beta2 = beta1, beta2
        self.epsilon = epsilon if epsilon is not None else (1e-8 if kind == 'adam' else 1e-10)
        self.slots = {}                      # variable op_name -> (slot0, slot1) device tensors
        self.step = 0                        # number of apply_gradients calls (adam's t)

    def lr_at(self, global_step):
        return float(self.learning_rate(global_step)) if callable(self.learning_rate) else float(self.learning_rate)

    def _slots_for(self, var):
        s = self.slots.get(var.op_name)
        if s is None:
            z = lambda: torch.zeros_like(var.tensor)
            if self.kind == 'sgd':
                s = (None, None)
            elif self.kind == 'momentum':
                s = (z(), None)
            elif self.kind == 'adam':
                s = (z(), z())
            else:                            # rmsprop: ms initialised to ones, mom to zeros (TF1)
                s = (torch.ones_like(var.tensor), z())
            self.slots[var.op_name] = s
        return s


# --------------------------------------------------------------------------------------------------------
# graph topology (queried from the C++ launch plan: one source of truth)
# --------------------------------------------------------------------------------------------------------
class _Topology(object):
    def __init__(self, class_num):
        L = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(L.y3_net_create(None, int(class_num), ctypes.byref(h)))
        self.layers = []
        ci = lambda: ctypes.c_int()
        for i in range(L.y3_net_num_layers(h)):
            k, s, cin, cout, bn = ci(), ci(), ci(), ci(), ci()
            _lib.check(L.y3_net_layer_info(h, i, *[ctypes.byref(v) for v in (k, s, cin, cout, bn)]))
            src, up, resid, dst, act = ci(), ci(), ci(), ci(), ci()
            _lib.check(L.y3_net_layer_graph(h, i, *[ctypes.byref(v) for v in (src, up, resid, dst, act)]))
            self.layers.append(dict(k=k.value, stride=s.value, cin=cin.value, cout=cout.value, bn=bool(bn.value),
                                    src=src.value, up=up.value, resid=resid.value, dst=dst.value, act=act.value))
        self.tensors = []
        for t in range(L.y3_net_num_tensors(h)):
            c, sd, ext = ci(), ci(), ci()
            _lib.check(L.y3_net_tensor_info(h, t, ctypes.byref(c), ctypes.byref(sd), ctypes.byref(ext)))
            self.tensors.append(dict(c=c.value, sdiv=sd.value, ext=ext.value))
        L.y3_net_destroy(h)


def _scratch(state, key, nbytes, device):
    t = state.setdefault('scratch', {}).get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        state['scratch'][key] = t
    return t


def _train_state(model):
    st = getattr(model, '_train', None)
    if st is None:
        st = dict(topo=_Topology(model.class_num), net=None)
        model._train = st
    return st


# y3_net dtype codes of the compute modes the train step accepts (model.NET_DTYPES without bf16 storage)
_TRAIN_DTYPES = {'f32': 0, 'f32_bf16x6': 2, 'f32_bf16x3': 3, 'f32_wino': 4}


def _train_net(model, ctx, dev=None):
    """The y3_net handle the train step runs on (one per model; its dtype follows model.compute_dtype at every call: the
    dtype only selects kernels, the state a forward leaves is the same in every mode)."""
    st = _train_state(model)
    L = _lib.lib()
    if st['net'] is None or st.get('net_ctx') != ctx.value:
        h = ctypes.c_void_p()
        _lib.check(L.y3_net_create(ctx, int(model.class_num), ctypes.byref(h)))
        st['net'], st['net_ctx'] = h, ctx.value
    mode = getattr(model, 'compute_dtype', 'f32')
    if mode not in _TRAIN_DTYPES:
        raise ValueError("training needs compute_dtype in %s (got %r)" % (sorted(_TRAIN_DTYPES), mode))
    if st.get('net_dtype') != mode:
        _lib.check(L.y3_net_set_dtype(st['net'], _TRAIN_DTYPES[mode]))
        st['net_dtype'] = mode
    # backward's second stream (include/yolo355.h: y3_net_train_set_wgrad_stream): on unless model.wgrad_stream is False
    # True: on a low-priority stream the library creates; False: off; a torch.cuda.Stream: on that stream
    want = getattr(model, 'wgrad_stream', False)
    handle = (want.cuda_stream if isinstance(want, torch.cuda.Stream) else 1) if want else None       # 1 = Y3_OWN_STREAM
    key = (handle, st['net'].value)
    if st.get('wgrad_key') != key:
        if isinstance(want, torch.cuda.Stream):
            st['side_stream'] = want        # (kept alive)
        _lib.check(L.y3_net_train_set_wgrad_stream(st['net'], ctypes.c_void_p(handle) if handle else None))
        st['wgrad_key'] = key
        st['ws_shape'] = None            # (the allocation sequence of backward changes with it)
    return st['net']


def _var_table(layer_vars, offsets=None, layer_ends=None):
    """ctypes array of y3_train_var for the 75 layers.  offsets: {op_name: element offset into the flat gradient buffer}
    for the trainable variables (None: every trainable variable, in gradient_layout's order - used to size the workspace);
    layer_ends: {layer index: offset just after that layer's gradients} (the `ready` hook's argument)."""
    if offsets is None:
        _, offsets, layer_ends, _ = gradient_layout(layer_vars)
    arr = (_lib.TrainVar * len(layer_vars))()
    keep = []
    off = lambda v: int(offsets.get(v.op_name, -1)) if v is not None else -1
    ptr = lambda v: v.tensor.data_ptr() if v is not None else None
    for i, (wvar, bnv, bias) in enumerate(layer_vars):
        g, b, mm, mv = bnv if bnv is not None else (None, None, None, None)
        arr[i] = _lib.TrainVar(ptr(wvar), ptr(g), ptr(b), ptr(mm), ptr(mv), ptr(bias), off(wvar), off(g), off(b), off(bias),
                               int((layer_ends or {}).get(i, -1)))
        keep.append([v.tensor for v in (wvar, g, b, mm, mv, bias) if v is not None])
    return arr, keep


def _opts(model):
    anc = np.ascontiguousarray(np.asarray(model.anchors, np.float32).reshape(9, 2))
    o = _lib.TrainOpts(float(model.batch_norm_decay), 1 if model.use_label_smooth else 0, 1 if model.use_focal_loss else 0,
                       anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return o, anc


# --------------------------------------------------------------------------------------------------------
# forward in training mode
# --------------------------------------------------------------------------------------------------------
def _prepare(model, x):
    """Everything a training forward needs before the library call: the train net in the model's mode, the variables (created
    on first use, in the reference's order), their table, the workspace (sized for every variable trainable: an upper bound
    for any update_vars subset), the options."""
    st = _train_state(model)
    topo = st['topo']
    L = _lib.lib()
    dev = x.device
    net = _train_net(model, fw.context(dev), dev)
    scope = fw.current_scope_name()
    layer_vars = model._ensure_variables(scope, [(l['k'], l['stride'], l['cin'], l['cout'], l['bn'])
                                                 for l in topo.layers])
    n, h, w, _ = x.shape
    key = tuple(t.tensor.data_ptr() for lv in layer_vars for t in ((lv[0],) + tuple(lv[1] or ()) + ((lv[2],) if lv[2] is not None else ())))
    if st.get('vars_key') != key:
        st['vars_all'], st['vars_keep'] = _var_table(layer_vars)
        st['vars_key'] = key
    ws_key = (n, h, w, str(dev), model.compute_dtype)      # the allocation sequence depends on the net's dtype too
    if st.get('ws_shape') != ws_key:
        need = L.y3_net_train_workspace_bytes(net, st['vars_all'], n, h, w)
        if need == 0:
            _lib.check(_lib.Y3_EINVAL)
        ws = st.get('ws')
        if ws is None or ws.numel() < need or ws.device != dev:
            st['ws'] = None              # (free the old one first)
            st['ws'] = torch.empty(need, dtype=torch.uint8, device=dev)
        st['ws_shape'] = ws_key
    st.update(layer_vars=layer_vars, batch=n, x=x, have_loss=False)
    return st, net, layer_vars


def _after_forward(model, st, fm_ptrs):
    """Bookkeeping after a training forward: the moving statistics changed in place; the feature maps are views of the
    workspace."""
    for wvar, bnv, bias in st['layer_vars']:
        if bnv is not None:        # updated in place: the folded inference parameters must be rebuilt
            bnv[2].touch()
            bnv[3].touch()
    ws, x = st['ws'], st['x']
    n, h, w, _ = x.shape
    det = 3 * (5 + model.class_num)
    fms = []
    for p, s in zip(fm_ptrs, (32, 16, 8)):
        o = p - ws.data_ptr()
        fms.append(ws[o:o + n * (h // s) * (w // s) * det * 4].view(torch.float32).view(n, h // s, w // s, det))
    st.update(fms=fms, fm_ptrs=list(fm_ptrs))
    return fms


def forward_train(model, x):
    """yolov3.forward(inputs, is_training=True): BN normalises with batch statistics in ALL 72 BN layers and
    updates the moving statistics (decay = model.batch_norm_decay); activations are kept for backward.
    ONE library call (y3_net_train_forward); the feature maps are views into the step's workspace."""
    st, net, layer_vars = _prepare(model, x)
    n, h, w, _ = x.shape
    opts, anc = _opts(model)
    fm = [ctypes.c_void_p() for _ in range(3)]
    _lib.check(_lib.lib().y3_net_train_forward(net, st['vars_all'], fw.ptr(x), n, h, w, ctypes.byref(opts), fw.ptr(st['ws']),
                                               ctypes.c_size_t(st['ws'].numel()), *[ctypes.byref(p) for p in fm]))
    fms = _after_forward(model, st, [p.value for p in fm])
    return fms[0], fms[1], fms[2]


# --------------------------------------------------------------------------------------------------------
# loss (model.py:192-365)
# --------------------------------------------------------------------------------------------------------
def _loss_one_scale(model, fm, y_true, anchors, loss4, accumulate, grad):
    L = _lib.lib()
    fm = fw.as_device_f32(fm)
    y_true = fw.as_device_f32(y_true)
    n, gh, gw, ch = fm.shape
    C = int(model.class_num)
    if ch != 3 * (5 + C):
        raise ValueError("feature map has %d channels, expected %d" % (ch, 3 * (5 + C)))
    if tuple(y_true.shape) != (n, gh, gw, 3, 6 + C):
        raise ValueError("y_true shape %s does not match the feature map (expected %s)" %
                         (tuple(y_true.shape), (n, gh, gw, 3, 6 + C)))
    if model.img_size is None:
        raise ValueError("loss needs img_size: call forward() first")
    st = _train_state(model)
    sc = _scratch(st, 'loss', L.y3_loss_scratch_bytes(n, gh, gw), fm.device)
    anc = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(3, 2))
    _lib.check(L.y3_loss_layer(fw.context(fm.device), fw.ptr(fm), fw.ptr(y_true), n, gh, gw, C,
                               int(model.img_size[0]), int(model.img_size[1]),
                               anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                               1 if model.use_label_smooth else 0, 1 if model.use_focal_loss else 0,
                               1 if accumulate else 0, fw.ptr(loss4), fw.ptr(grad), int(grad.shape[-1]),
                               fw.ptr(sc), ctypes.c_size_t(sc.numel())))


def loss_layer(model, feature_map_i, y_true, anchors):
    """reference model.py:192-304: returns (xy_loss, wh_loss, conf_loss, class_loss) device scalars."""
    fm = fw.as_device_f32(feature_map_i)
    loss4 = torch.zeros(4, dtype=torch.float32, device=fm.device)
    grad = torch.zeros(tuple(fm.shape[:3]) + (fm.shape[3],), dtype=torch.float32, device=fm.device)
    _loss_one_scale(model, fm, y_true, anchors, loss4, False, grad)
    return loss4[0], loss4[1], loss4[2], loss4[3]


def compute_loss(model, y_pred, y_true):
    """reference model.py:348-365: [total, xy, wh, conf, class] over the three scales (anchors [6:9], [3:6],
    [0:3]).  When y_pred are the feature maps of the last forward(is_training=True) this is ONE library call
    (y3_net_train_loss) that also leaves d(total)/d(feature_map_i) in the step's workspace for the backward pass; any other
    feature maps (validation: forward(is_training=False)) take the per-scale entry point."""
    st = _train_state(model)
    L = _lib.lib()
    if st.get('fm_ptrs') is not None and [fw.as_device_f32(f).data_ptr() for f in y_pred] == st['fm_ptrs']:
        dev = st['fms'][0].device
        yt = [fw.as_device_f32(y) for y in y_true]
        for f, y in zip(st['fms'], yt):
            if tuple(y.shape) != tuple(f.shape[:3]) + (3, 6 + int(model.class_num)):
                raise ValueError("y_true shape %s does not match the feature map %s" % (tuple(y.shape), tuple(f.shape)))
        opts, anc = _opts(model)
        loss5 = torch.empty(5, dtype=torch.float32, device=dev)
        _lib.check(L.y3_net_train_loss(st['net'], fw.ptr(yt[0]), fw.ptr(yt[1]), fw.ptr(yt[2]), ctypes.byref(opts),
                                       fw.ptr(loss5)))
        st['have_loss'] = True
        return [loss5[0], loss5[1], loss5[2], loss5[3], loss5[4]]
    anchors = np.asarray(model.anchors, np.float32).reshape(9, 2)
    groups = [anchors[6:9], anchors[3:6], anchors[0:3]]
    dev = fw.as_device_f32(y_pred[0]).device
    loss4 = torch.zeros(4, dtype=torch.float32, device=dev)
    det_pad = ((3 * (5 + model.class_num) + 31) // 32) * 32
    grads = []
    for i in range(3):
        fm = fw.as_device_f32(y_pred[i])
        g = torch.zeros(tuple(fm.shape[:3]) + (det_pad,), dtype=torch.float32, device=dev)   # pad lanes stay 0
        _loss_one_scale(model, fm, y_true[i], groups[i], loss4, i > 0, g)
        grads.append(g)
    st['fm_grads'] = grads       # (test hook: d total / d feature_map_i of this stand-alone evaluation)
    total = loss4.sum()
    return [total, loss4[0], loss4[1], loss4[2], loss4[3]]


def box_iou(pred_boxes, valid_true_boxes):
    """reference model.py:307-345: pred_boxes [g,g,3,4] (cx,cy,w,h), valid_true_boxes [V,4] -> iou [g,g,3,V].
    (Inside compute_loss the same arithmetic is fused into the loss kernel.)"""
    pb = fw.as_device_f32(pred_boxes)
    tb = fw.as_device_f32(valid_true_boxes).reshape(-1, 4)
    if pb.shape[-1] != 4:
        raise ValueError("pred_boxes must end in 4 (cx, cy, w, h)")
    v = int(tb.shape[0])
    out = torch.empty(tuple(pb.shape[:-1]) + (v,), dtype=torch.float32, device=pb.device)
    if v == 0 or pb.numel() == 0:
        return out
    _lib.check(_lib.lib().y3_box_iou(fw.context(pb.device), fw.ptr(pb), pb.numel() // 4, fw.ptr(tb), v, fw.ptr(out)))
    return out


# --------------------------------------------------------------------------------------------------------
# backward + update
# --------------------------------------------------------------------------------------------------------
def gradient_layout(layer_vars, trainable=lambda v: v.trainable):
    """Layout of the flat gradient buffer: the variables in the order backward PRODUCES their gradients (last layer
    first; inside a layer gamma, beta / bias, then the kernel), every view 16-byte aligned.  Pure host arithmetic over
    objects with .op_name / .shape (no device needed).
    Returns (order, offsets {op_name: element offset}, layer_ends {layer index: element offset just after that layer's
    gradients} (layers with a trainable variable only; increasing as the layer index decreases), total elements)."""
    order, offs, ends, total = [], {}, {}, 0
    layer_vars = list(layer_vars)
    for li in range(len(layer_vars) - 1, -1, -1):
        wvar, bnv, bias = layer_vars[li]
        vs = (list(bnv[:2]) if bnv is not None else [bias]) + [wvar]
        vs = [v for v in vs if trainable(v)]
        for v in vs:
            numel = 1
            for d in v.shape:
                numel *= int(d)
            offs[v.op_name] = total
            total += (numel + 3) // 4 * 4
            order.append(v)
        if vs:
            ends[li] = total
    return order, offs, ends, total


class Trainer(object):
    """One object = the train op of train.py:105-115: gradients of loss[0] + l2_loss w.r.t. update_vars,
    per-tensor clip_by_norm(100), optimizer.apply_gradients; with torch.distributed initialised, gradients
    are summed over ranks — RCCL all-reduce of the flat gradient buffer in buckets, issued in the order backward
    produces them and overlapped with the rest of backward (distributed.GradientExchange) — and averaged (the 1/world
    factor is folded into the clip/update kernel) before clipping."""

    def __init__(self, model, optimizer, update_vars=None, clip_norm=CLIP_NORM, process_group=None,
                 global_step=0.0, bucket_bytes=distributed.DEFAULT_BUCKET_BYTES, wgrad_stream='auto'):
        self.model, self.opt = model, optimizer
        # backward's weight gradients on a second stream (include/yolo355.h: y3_net_train_set_wgrad_stream): True / False /
        # a torch.cuda.Stream / 'auto'.  Every result is bit-identical either way; only the step time differs - and by how
        # much depends on which hardware queue the second stream lands on (with many streams alive in the process it can
        # share the main stream's: 96 ms per bs=64 step instead of 79.6 against 81 on one stream,
        # profiles/r06_wgrad_stream_ab.txt).  'auto' therefore MEASURES: steps 3-4 with it, steps 6-7 without (hipEvents
        # around the step, one host synchronisation each), and keeps the faster; `wgrad_choice` tells which.
        self.wgrad_stream = wgrad_stream
        self.wgrad_choice = None if wgrad_stream == 'auto' else wgrad_stream
        self._calib = dict(step=0, on=[], off=[]) if wgrad_stream == 'auto' else None
        self.update_names = None if update_vars is None else set(v.op_name for v in update_vars)
        self.clip_norm = float(clip_norm)
        self.pg = process_group
        self.bucket_bytes = int(bucket_bytes)
        self.global_step = float(global_step)
        self.flat = None
        self.views = None
        self.exchange = None
        self.capture = None      # test hook: a list that backward() fills with each BN layer's (z, stats)

    def _trainable(self, var):
        return var.trainable and (self.update_names is None or var.op_name in self.update_names)

    def _alloc_grads(self, layer_vars, dev):
        if self.flat is not None:
            return
        order, offs, ends, total = gradient_layout(layer_vars, self._trainable)
        self.flat = torch.zeros(max(total, 4), dtype=torch.float32, device=dev)
        self.views = {v.op_name: self.flat[offs[v.op_name]:offs[v.op_name] + v.tensor.numel()].view(v.tensor.shape)
                      for v in order}
        self.order = order
        self.offsets = offs
        self.layer_ends = ends
        self.exchange = distributed.GradientExchange(self.flat, sorted(ends.values()) or [total], self.pg,
                                                     self.bucket_bytes)

    def _vars(self, layer_vars, dev):
        self._alloc_grads(layer_vars, dev)
        key = tuple(v.tensor.data_ptr() for v in self.order) + tuple(
            t.tensor.data_ptr() for lv in layer_vars for t in (lv[1][2:] if lv[1] is not None else ()))
        if getattr(self, '_vars_key', None) != key:
            self._vars_arr, self._vars_keep = _var_table(layer_vars, self.offsets, self.layer_ends)
            self._vars_key = key
        return self._vars_arr

    def _ready_cb(self):
        if getattr(self, '_cb', None) is None:
            self._cb = _lib.GradReadyFn(lambda user, edge: self.exchange.ready(int(edge)))
        return self._cb

    def _fill_capture(self, st):
        """Test hook: views of the tensors that fix each BN layer's LeakyReLU branch (z and the folded scale / shift)."""
        L = _lib.lib()
        topo, ws, n = st['topo'], st['ws'], st['batch']
        h, w = st['x'].shape[1:3]
        for i in range(len(topo.layers) - 1, -1, -1):
            l = topo.layers[i]
            zo, so = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(L.y3_net_train_saved(st['net'], i, ctypes.byref(zo), ctypes.byref(so)))
            if not l['bn']:
                self.capture.append(dict(layer=i, z=None, stats=None))
                continue
            sd = topo.tensors[l['dst']]['sdiv']
            cout = l['cout']
            z = ws[zo.value:zo.value + n * (h // sd) * (w // sd) * cout * 4].view(torch.float32).view(n, h // sd, w // sd, cout)
            stats = ws[so.value:so.value + 4 * cout * 4].view(torch.float32).view(4, cout)
            self.capture.append(dict(layer=i, z=z, stats=stats))

    def backward(self):
        """train.py:112 compute_gradients: ONE library call (y3_net_train_backward) walks the graph backwards and writes every
        trainable variable's gradient into the flat buffer; the `ready` hook issues the gradient buckets as they complete."""
        st = _train_state(self.model)
        if not st.get('have_loss') or st.get('fms') is None:
            raise RuntimeError('backward needs forward(is_training=True) and compute_loss first')
        dev = st['fms'][0].device
        net = _train_net(self.model, fw.context(dev), dev)         # (the mode may have changed since the forward)
        arr = self._vars(st['layer_vars'], dev)
        self.exchange.begin()
        _lib.check(_lib.lib().y3_net_train_backward(net, arr, fw.ptr(self.flat), self._ready_cb(), None))
        if self.capture is not None:
            self._fill_capture(st)

    def _param_descs(self):
        """Host array of y3_param_desc for self.order (rebuilt when a variable's storage or a slot changed)."""
        sig = tuple((v.tensor.data_ptr(),) + tuple(0 if s is None else s.data_ptr() for s in self.opt._slots_for(v))
                    for v in self.order)
        if getattr(self, '_descs_sig', None) != sig:
            arr = (_lib.ParamDesc * len(self.order))()
            keep = []
            for i, v in enumerate(self.order):
                g = self.views[v.op_name]
                s0, s1 = self.opt._slots_for(v)
                wd = float(self.model.weight_decay) if v.op_name.endswith('/weights') else 0.0
                arr[i] = _lib.ParamDesc(v.tensor.data_ptr(), g.data_ptr(), 0 if s0 is None else s0.data_ptr(),
                                        0 if s1 is None else s1.data_ptr(), v.tensor.numel(), wd, 0)
                keep.append((v.tensor, s0, s1))
            self._descs, self._descs_keep, self._descs_sig = arr, keep, sig
        return self._descs, self._descs_keep

    def apply_gradients(self):
        """all-reduce (mean over ranks) -> + weight_decay*w on conv kernels -> clip_by_norm -> update."""
        L = _lib.lib()
        dev = self.flat.device
        ctx = fw.context(dev)
        self.exchange.finish()      # buckets not yet issued by backward + join the asynchronous ones
        world = self.exchange.world
        self.opt.step += 1
        lr = self.opt.lr_at(self.global_step)
        kind = Optimizer.KINDS.index(self.opt.kind)
        if self.opt.kind == 'adam':
            t = self.opt.step
            lr = lr * np.sqrt(1.0 - self.opt.beta2 ** t) / (1.0 - self.opt.beta1 ** t)
        decay = self.opt.beta1 if self.opt.kind == 'adam' else self.opt.decay
        st = _train_state(self.model)
        if os.environ.get('Y3_TRAIN_PER_TENSOR_UPDATE') == '1':     # experiment hook: one y3_clip_update per variable
            sc = _scratch(st, 'opt', L.y3_optimizer_scratch_bytes(), dev)
            for v in self.order:
                s0, s1 = self.opt._slots_for(v)
                wd = float(self.model.weight_decay) if v.op_name.endswith('/weights') else 0.0
                _lib.check(L.y3_clip_update(ctx, kind, fw.ptr(v.tensor), fw.ptr(self.views[v.op_name]), fw.ptr(s0),
                                            fw.ptr(s1), v.tensor.numel(), ctypes.c_float(wd), ctypes.c_float(1.0 / world),
                                            ctypes.c_float(self.clip_norm), ctypes.c_float(lr),
                                            ctypes.c_float(self.opt.momentum), ctypes.c_float(decay),
                                            ctypes.c_float(self.opt.beta2), ctypes.c_float(self.opt.epsilon), fw.ptr(sc)))
                v.touch()
            self.global_step += 1.0
            return
        # ONE multi-tensor call (three launches) for every (gradient, variable) pair of train.py:112-115
        descs, keep = self._param_descs()
        nbytes = L.y3_clip_update_multi_scratch_bytes(descs, len(self.order))
        sc = _scratch(st, 'opt_multi', nbytes, dev)
        _lib.check(L.y3_clip_update_multi(ctx, kind, descs, len(self.order), ctypes.c_float(1.0 / world),
                                          ctypes.c_float(self.clip_norm), ctypes.c_float(lr),
                                          ctypes.c_float(self.opt.momentum), ctypes.c_float(decay),
                                          ctypes.c_float(self.opt.beta2), ctypes.c_float(self.opt.epsilon),
                                          fw.ptr(sc), ctypes.c_size_t(sc.numel())))
        for v in self.order:
            v.touch()
        self.global_step += 1.0

    def step(self, images, y_true):
        """One training step on a batch; returns [total, xy, wh, conf, class] (device scalars):
        forward(is_training=True) -> compute_loss -> backward in ONE library call (y3_net_train_step), then the
        all-reduce join, L2, clip and update (y3_clip_update_multi)."""
        x = fw.as_device_f32(images)
        self.model.img_size = [int(x.shape[1]), int(x.shape[2])]
        timing = None
        if self._calib is not None:
            k = self._calib['step']
            self.model.wgrad_stream = k < 4                   # steps 0-3 with the second stream, 4-6 without
            if k in (2, 3, 5, 6):
                timing = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                timing[0].record(torch.cuda.current_stream(x.device))
        else:
            self.model.wgrad_stream = self.wgrad_choice
        st, net, layer_vars = _prepare(self.model, x)
        n, h, w, _ = x.shape
        yt = [fw.as_device_f32(y) for y in y_true]
        for y, s in zip(yt, (32, 16, 8)):
            if tuple(y.shape) != (n, h // s, w // s, 3, 6 + int(self.model.class_num)):
                raise ValueError("y_true shape %s does not match the input size" % (tuple(y.shape),))
        arr = self._vars(layer_vars, x.device)
        opts, anc = _opts(self.model)
        loss5 = torch.empty(5, dtype=torch.float32, device=x.device)
        self.exchange.begin()
        _lib.check(_lib.lib().y3_net_train_step(net, arr, fw.ptr(x), n, h, w, fw.ptr(yt[0]), fw.ptr(yt[1]), fw.ptr(yt[2]),
                                                ctypes.byref(opts), fw.ptr(self.flat), fw.ptr(st['ws']),
                                                ctypes.c_size_t(st['ws'].numel()), fw.ptr(loss5), self._ready_cb(), None))
        st.update(have_loss=True, fms=None, fm_ptrs=None)
        for wvar, bnv, bias in layer_vars:
            if bnv is not None:
                bnv[2].touch()
                bnv[3].touch()
        if self.capture is not None:
            self._fill_capture(st)
        self.apply_gradients()
        if self._calib is not None:
            c = self._calib
            if timing is not None:
                timing[1].record(torch.cuda.current_stream(x.device))
                timing[1].synchronize()
                (c['on'] if c['step'] < 4 else c['off']).append(timing[0].elapsed_time(timing[1]))
            c['step'] += 1
            if c['step'] == 7:
                self.wgrad_choice = min(c['on']) < 0.995 * min(c['off'])
                self.wgrad_calibration = dict(ms_with=min(c['on']), ms_without=min(c['off']))
                self._calib = None
        return [loss5[0], loss5[1], loss5[2], loss5[3], loss5[4]]
