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
from . import engine
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
        st = dict(topo=_Topology(model.class_num), packed={}, consts={})
        model._train = st
    return st


def _consts(st, c, device):
    k = (c, str(device))
    v = st['consts'].get(k)
    if v is None:
        v = (torch.ones(c, device=device), torch.zeros(c, device=device))
        st['consts'][k] = v
    return v


def _planes(model):
    """3 / 2 when the model computes its convs on the bf16 matrix pipe (compute_dtype 'f32_bf16x6' / 'f32_bf16x3'):
    the training forward and the stride-1 data gradients then use the split kernels too (weight gradients and the
    stride-2 data gradients stay on the exact fp32 kernels)."""
    return engine.SPLIT_PLANES.get(getattr(model, 'compute_dtype', 'f32'), 0)


def _packed_weights(st, wvar, planes=0):
    """[tap][Cout][Cin] re-pack of the HWIO variable for the forward conv (or its split-plane packing), refreshed
    when the variable changes."""
    key = wvar.op_name + ('#%d' % planes if planes else '')
    hit = st['packed'].get(key)
    if hit is not None and hit[0] == wvar.version:
        return hit[1]
    k, _, cin, cout = wvar.shape
    if cin == 3:
        wp = wvar.tensor
    elif planes:
        wp = hit[1] if hit is not None else torch.empty(planes * k * k * cout * cin, dtype=torch.bfloat16,
                                                        device=wvar.tensor.device)
        _lib.check(_lib.lib().y3_pack_conv_weights_split(fw.context(wvar.tensor.device), fw.ptr(wvar.tensor), k, cin,
                                                         cout, planes, fw.ptr(wp)))
    else:
        wp = hit[1] if hit is not None else torch.empty(k * k * cout * cin, dtype=torch.float32,
                                                        device=wvar.tensor.device)
        _lib.check(_lib.lib().y3_pack_conv_weights(fw.context(wvar.tensor.device), fw.ptr(wvar.tensor), k, cin, cout,
                                                   fw.ptr(wp)))
    st['packed'][key] = (wvar.version, wp)
    return wp


# --------------------------------------------------------------------------------------------------------
# forward in training mode
# --------------------------------------------------------------------------------------------------------
def forward_train(model, x):
    """yolov3.forward(inputs, is_training=True): BN normalises with batch statistics in ALL 72 BN layers and
    updates the moving statistics (decay = model.batch_norm_decay); activations are kept for backward."""
    st = _train_state(model)
    topo = st['topo']
    L = _lib.lib()
    dev = x.device
    ctx = fw.context(dev)
    scope = fw.current_scope_name()
    layer_vars = model._ensure_variables(scope, [(l['k'], l['stride'], l['cin'], l['cout'], l['bn'])
                                                 for l in topo.layers])
    n, h, w, _ = x.shape
    tens = {0: x}
    saved = []
    for i, l in enumerate(topo.layers):
        wvar, bnv, bias = layer_vars[i]
        xin = tens[l['src']]
        if l['up'] >= 0:   # training materialises concat([upsample(up), route]) (model.py:61-62,71-72)
            upt = engine.upsample_nearest(tens[l['up']], xin.shape[1], xin.shape[2])
            xin = engine.concat_channels(upt, xin)
        cout = l['cout']
        ones, zeros = _consts(st, cout, dev)
        planes = _planes(model) if l['cin'] != 3 else 0
        wino = (getattr(model, 'compute_dtype', 'f32') == 'f32_wino' and
                engine.wino_eligible(l['k'], l['stride'], int(xin.shape[3]), cout))
        rec = dict(xin=xin)
        # exact-fp32 BN layers: the conv also returns the column sums of z and z^2 per row block of its output (taken in
        # its epilogue), which saves the statistics' own pass over z; the stem and the split-precision kernels keep it
        desc = _lib.ConvDesc(n, int(xin.shape[1]), int(xin.shape[2]), int(xin.shape[3]), 0, cout, l['k'], l['stride'], 0)
        nblk = L.y3_conv_stats_blocks(ctypes.byref(desc), 1 if wino else 0) if (l['bn'] and not planes) else 0
        part = torch.empty((nblk, 2, cout), dtype=torch.float32, device=dev) if nblk else None
        if wino:        # Winograd forward for the stride-1 3x3 convs
            key = wvar.op_name + '#wino'
            hit = st['packed'].get(key)
            if hit is None or hit[0] != wvar.version:
                hit = (wvar.version, engine.pack_wino(wvar.tensor))
                st['packed'][key] = hit
            z = engine.conv2d_fwd_wino(xin, hit[1], ones, zeros, cout, False, stats=part)
            wp = None
        else:
            wp = _packed_weights(st, wvar, planes)
        if l['bn']:
            if not wino:
                z = engine.conv2d_fwd(xin, wp, ones, zeros, l['k'], l['stride'], cout, False, planes=planes, stats=part)
            rows = z.numel() // cout
            stats = torch.empty((4, cout), dtype=torch.float32, device=dev)   # mean, inv_std, scale, shift
            gamma, beta, mmean, mvar = bnv
            if part is not None:
                _lib.check(L.y3_bn_train_stats_partials(
                    ctx, fw.ptr(part), nblk, rows, cout, fw.ptr(gamma.tensor), fw.ptr(beta.tensor),
                    ctypes.c_float(BN_EPS), ctypes.c_float(model.batch_norm_decay), fw.ptr(stats[0]), fw.ptr(stats[1]),
                    fw.ptr(stats[2]), fw.ptr(stats[3]), fw.ptr(mmean.tensor), fw.ptr(mvar.tensor)))
            else:
                sc = _scratch(st, 'reduce', L.y3_reduce_scratch_bytes(cout), dev)
                _lib.check(L.y3_bn_train_stats(ctx, fw.ptr(z), rows, cout, fw.ptr(gamma.tensor), fw.ptr(beta.tensor),
                                               ctypes.c_float(BN_EPS), ctypes.c_float(model.batch_norm_decay),
                                               fw.ptr(stats[0]), fw.ptr(stats[1]), fw.ptr(stats[2]), fw.ptr(stats[3]),
                                               fw.ptr(mmean.tensor), fw.ptr(mvar.tensor), fw.ptr(sc)))
            mmean.touch()      # updated in place: the folded inference parameters must be rebuilt
            mvar.touch()
            y = torch.empty_like(z)
            resid = tens[l['resid']] if l['resid'] >= 0 else None
            _lib.check(L.y3_bn_apply_fwd(ctx, fw.ptr(z), fw.ptr(stats[2]), fw.ptr(stats[3]), fw.ptr(resid), rows,
                                         cout, 1, fw.ptr(y)))
            rec.update(z=z, stats=stats)
        else:              # detection conv: bias, linear (model.py:55-57)
            y = engine.conv2d_fwd(xin, wp, ones, bias.tensor, l['k'], l['stride'], cout, False, planes=planes)
        tens[l['dst']] = y
        saved.append(rec)
    fms = [tens[t] for t in range(len(topo.tensors)) if topo.tensors[t]['ext'] >= 0]
    fms.sort(key=lambda f: f.shape[1])                          # ext slots 0,1,2 = 13-, 26-, 52-grid
    st.update(saved=saved, tens=tens, layer_vars=layer_vars, fm_grads=None, batch=n)
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
    [0:3]).  Also leaves d(total)/d(feature_map_i) in the model's train state for the backward pass."""
    anchors = np.asarray(model.anchors, np.float32).reshape(9, 2)
    groups = [anchors[6:9], anchors[3:6], anchors[0:3]]
    st = _train_state(model)
    dev = fw.as_device_f32(y_pred[0]).device
    loss4 = torch.zeros(4, dtype=torch.float32, device=dev)
    det_pad = ((3 * (5 + model.class_num) + 31) // 32) * 32
    grads = []
    for i in range(3):
        fm = fw.as_device_f32(y_pred[i])
        g = torch.zeros(tuple(fm.shape[:3]) + (det_pad,), dtype=torch.float32, device=dev)   # pad lanes stay 0
        _loss_one_scale(model, fm, y_true[i], groups[i], loss4, i > 0, g)
        grads.append(g)
    st['fm_grads'] = grads
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
                 global_step=0.0, bucket_bytes=distributed.DEFAULT_BUCKET_BYTES):
        self.model, self.opt = model, optimizer
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

    def backward(self):
        model = self.model
        st = _train_state(model)
        if st.get('fm_grads') is None or st.get('saved') is None:
            raise RuntimeError('backward needs forward(is_training=True) and compute_loss first')
        topo, saved, tens, layer_vars = st['topo'], st['saved'], st['tens'], st['layer_vars']
        L = _lib.lib()
        dev = tens[0].device
        ctx = fw.context(dev)
        self._alloc_grads(layer_vars, dev)
        n = st['batch']
        # earliest layer holding a trainable variable: gradients need not flow below it
        first = None
        for i, (wvar, bnv, bias) in enumerate(layer_vars):
            vs = [wvar] + (list(bnv[:2]) if bnv is not None else [bias])
            if any(self._trainable(v) for v in vs):
                first = i
                break
        if first is None:
            return
        self.exchange.begin()
        cap = self.capture
        needs = lambda t: t > 0 and (t - 1) >= first         # tensor t is produced by layer t-1
        grads, have = {}, set()
        fm_ids = sorted([t for t in range(len(topo.tensors)) if topo.tensors[t]['ext'] >= 0],
                        key=lambda t: topo.tensors[t]['ext'])
        for t, g in zip(fm_ids, st['fm_grads']):
            grads[t] = g
            have.add(t)
        # stream-K scratch of the data-gradient convs: the library's size for any 3x3 conv with Cout >= 128
        sk_desc = _lib.ConvDesc(1, 8, 8, 128, 0, 128, 3, 1, 0)
        sk_ws = _scratch(st, 'streamk', int(L.y3_conv_workspace_bytes(ctypes.byref(sk_desc))), dev)

        def accumulate_into(t, src, src_channels, offset, c):
            rows = src.numel() // src_channels
            if t not in have:
                grads[t] = torch.empty(tuple(tens[t].shape), dtype=torch.float32, device=dev)
            _lib.check(L.y3_slice_accumulate(ctx, fw.ptr(src), src_channels, offset, rows, c, 1 if t in have else 0,
                                             fw.ptr(grads[t])))
            have.add(t)

        for i in range(len(topo.layers) - 1, first - 1, -1):
            l = topo.layers[i]
            wvar, bnv, bias = layer_vars[i]
            rec = saved[i]
            dst = l['dst']
            if dst not in have:
                continue
            dy = grads[dst]
            cout = l['cout']
            xin = rec['xin']
            if cap is not None:      # test hook: the tensors that fix this layer's LeakyReLU branch (z, folded scale/shift)
                cap.append(dict(layer=i, z=rec.get('z'), stats=rec.get('stats')))
            if l['bn']:
                rows = dy.numel() // cout
                dz_out = dy                       # BN backward runs in place ...
                if l['resid'] >= 0 and needs(l['resid']):
                    if l['resid'] in have:
                        accumulate_into(l['resid'], dy, cout, 0, cout)
                    else:
                        # ... unless dy is also the first contribution to the shortcut's gradient (res_block: net +
                        # shortcut, utils/layer_utils.py:30): then dy itself becomes that gradient (no copy) and the
                        # BN backward writes dz elsewhere
                        grads[l['resid']] = dy
                        have.add(l['resid'])
                        dz_out = torch.empty_like(dy)
                gamma, beta = bnv[0], bnv[1]
                stats = rec['stats']
                tmp = torch.empty((2, cout), dtype=torch.float32, device=dev)
                dgam = self.views.get(gamma.op_name)
                dbet = self.views.get(beta.op_name)
                sc = _scratch(st, 'bnbwd', L.y3_bn_bwd_scratch_bytes(cout), dev)
                _lib.check(L.y3_bn_train_bwd(ctx, fw.ptr(rec['z']), fw.ptr(dy), fw.ptr(gamma.tensor),
                                             fw.ptr(stats[2]), fw.ptr(stats[3]), fw.ptr(stats[0]), fw.ptr(stats[1]),
                                             rows, cout, fw.ptr(dgam if dgam is not None else tmp[0]),
                                             fw.ptr(dbet if dbet is not None else tmp[1]), fw.ptr(dz_out), fw.ptr(sc)))
                dz, dz_stride, w_d = dz_out, cout, wvar.tensor
            else:
                dz_stride = int(dy.shape[-1])
                rows = dy.numel() // dz_stride
                if self._trainable(bias):
                    tmp = torch.empty(dz_stride, dtype=torch.float32, device=dev)
                    sc = _scratch(st, 'bias', 1024 * dz_stride * 4, dev)
                    _lib.check(L.y3_bias_grad(ctx, fw.ptr(dy), rows, dz_stride, fw.ptr(tmp), fw.ptr(sc)))
                    self.views[bias.op_name].copy_(tmp[:cout])
                dz = dy
                # the data gradient reads the kernel as [k*k][cin][dz_stride]: zero-extend its last axis
                k, _, cin, _ = wvar.shape
                w_d = torch.empty(k * k * cin * dz_stride, dtype=torch.float32, device=dev)
                _lib.check(L.y3_pad_channels(ctx, fw.ptr(wvar.tensor), cout, k * k * cin, dz_stride, fw.ptr(w_d)))
            d = _lib.ConvDesc(n, int(xin.shape[1]), int(xin.shape[2]), int(xin.shape[3]), 0, cout, l['k'],
                              l['stride'], 0)
            if self._trainable(wvar):
                # compute_dtype 'f32_wino': the stride-1 3x3 kernels' gradients in Winograd form too (16/36 of the MFMA work)
                if (getattr(model, 'compute_dtype', 'f32') == 'f32_wino' and
                        L.y3_conv_wgrad_wino_eligible(ctypes.byref(d)) == 1):
                    sc = _scratch(st, 'wgrad_wino', L.y3_conv_wgrad_wino_scratch_bytes(ctypes.byref(d)), dev)
                    _lib.check(L.y3_conv_wgrad_wino(ctx, ctypes.byref(d), fw.ptr(xin), fw.ptr(dz), dz_stride,
                                                    fw.ptr(self.views[wvar.op_name]), fw.ptr(sc),
                                                    ctypes.c_size_t(sc.numel())))
                else:
                    wsb = L.y3_conv_wgrad_scratch_bytes(ctypes.byref(d))
                    sc = _scratch(st, 'wgrad', wsb, dev)
                    _lib.check(L.y3_conv_wgrad(ctx, ctypes.byref(d), fw.ptr(xin), fw.ptr(dz), dz_stride,
                                               fw.ptr(self.views[wvar.op_name]), fw.ptr(sc),
                                               ctypes.c_size_t(sc.numel())))
            if i in self.layer_ends:      # this layer's gradients are complete: reduce every bucket below its edge
                self.exchange.ready(self.layer_ends[i])
            src, up = l['src'], l['up']
            need_src = needs(src)
            need_up = up >= 0 and needs(up)
            if not (need_src or need_up):
                continue
            cin = int(xin.shape[3])
            ones, zeros = _consts(st, cin, dev)
            planes = _planes(model) if (l['stride'] == 1 and cin % 4 == 0 and dz_stride % 32 == 0) else 0
            # compute_dtype 'f32_wino': the data gradient of a stride-1 3x3 conv is itself a stride-1 3x3 SAME conv
            # (flipped, channel-swapped kernel), so it runs on the Winograd kernel too
            wino_d = (getattr(model, 'compute_dtype', 'f32') == 'f32_wino' and up < 0 and
                      engine.wino_eligible(l['k'], l['stride'], dz_stride, cin))
            if wino_d:
                w_dw = _scratch(st, 'wwino_d', 16 * cin * dz_stride * 4, dev)
                _lib.check(L.y3_pack_conv_weights_wino_dgrad(ctx, fw.ptr(w_d), cin, dz_stride, fw.ptr(w_dw)))
                gdesc = _lib.ConvDesc(n, int(xin.shape[1]), int(xin.shape[2]), dz_stride, 0, cin, 3, 1, 0)
                wk_ws = _scratch(st, 'streamk_wino', int(L.y3_conv_wino_workspace_bytes(ctypes.byref(gdesc))), dev)
            if planes:
                k = l['k']
                w_ds = _scratch(st, 'wsplit_d', planes * k * k * cin * dz_stride * 2, dev)
                _lib.check(L.y3_pack_conv_weights_split_dgrad(ctx, fw.ptr(w_d), k, cin, dz_stride, planes,
                                                              fw.ptr(w_ds)))

            def dgrad(accumulate, dx):
                if wino_d:
                    _lib.check(L.y3_conv2d_dgrad_wino(ctx, ctypes.byref(d), fw.ptr(dz), dz_stride, fw.ptr(w_dw),
                                                      fw.ptr(ones), fw.ptr(zeros), accumulate, fw.ptr(dx),
                                                      fw.ptr(wk_ws), ctypes.c_size_t(wk_ws.numel())))
                elif planes:
                    _lib.check(L.y3_conv2d_dgrad_split(ctx, ctypes.byref(d), planes, fw.ptr(dz), dz_stride,
                                                       fw.ptr(w_ds), fw.ptr(ones), fw.ptr(zeros), accumulate,
                                                       fw.ptr(dx), fw.ptr(sk_ws), ctypes.c_size_t(sk_ws.numel())))
                else:
                    _lib.check(L.y3_conv2d_dgrad(ctx, ctypes.byref(d), fw.ptr(dz), dz_stride, fw.ptr(w_d),
                                                 fw.ptr(ones), fw.ptr(zeros), accumulate, fw.ptr(dx), fw.ptr(sk_ws),
                                                 ctypes.c_size_t(sk_ws.numel())))
            if up >= 0:
                dcat = torch.empty(tuple(xin.shape), dtype=torch.float32, device=dev)
                dgrad(0, dcat)
                cu = topo.tensors[up]['c']
                if need_up:
                    ut = tens[up]
                    if up not in have:
                        grads[up] = torch.empty(tuple(ut.shape), dtype=torch.float32, device=dev)
                    _lib.check(L.y3_upsample2x_bwd(ctx, fw.ptr(dcat), cin, n, int(ut.shape[1]), int(ut.shape[2]), cu,
                                                   1 if up in have else 0, fw.ptr(grads[up])))
                    have.add(up)
                if need_src:
                    accumulate_into(src, dcat, cin, cu, cin - cu)
            else:
                if src not in have:
                    grads[src] = torch.empty(tuple(tens[src].shape), dtype=torch.float32, device=dev)
                dgrad(1 if src in have else 0, grads[src])
                have.add(src)
            grads.pop(dst, None)      # free the consumed gradient
        st['saved'] = None
        st['fm_grads'] = None

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
        """One training step on a batch; returns [total, xy, wh, conf, class] (device scalars)."""
        fms = self.model.forward(images, is_training=True)
        loss = compute_loss(self.model, fms, y_true)
        self.backward()
        self.apply_gradients()
        return loss
