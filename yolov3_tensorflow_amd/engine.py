"""Host-side glue between named Variables and the C ABI: parameter preparation (re-pack + BN fold, cached
per variable version) and thin Python wrappers over the per-op entry points."""
import ctypes

import torch

from . import _lib
from . import framework as fw

BN_EPS = 1e-5  # model.py:37

_param_cache = {}  # weights op_name -> (version key, w_dev, scale, shift)


def prepare_conv_params_bf16(w_var, bn_vars=None, bias_var=None):
    """As prepare_conv_params, with the kernel packed to bf16 (the 3->32 stem keeps its fp32 HWIO kernel)."""
    w32, scale, shift = prepare_conv_params(w_var, bn_vars=bn_vars, bias_var=bias_var)
    k, _, cin, cout = w_var.shape
    if cin == 3:
        return w32, scale, shift
    key = w_var.op_name + '#bf16'
    hit = _param_cache.get(key)
    if hit is not None and hit[0] == w_var.version:
        return hit[1], scale, shift
    wb = torch.empty(k * k * cout * cin, dtype=torch.bfloat16, device=w_var.tensor.device)
    _lib.check(_lib.lib().y3_pack_conv_weights_bf16(fw.context(), fw.ptr(w_var.tensor), k, cin, cout, fw.ptr(wb)))
    _param_cache[key] = (w_var.version, wb)
    return wb, scale, shift


SPLIT_PLANES = {'f32_bf16x6': 3, 'f32_bf16x3': 2}


def prepare_conv_params_split(w_var, bn_vars=None, bias_var=None, planes=3):
    """As prepare_conv_params, with the kernel pre-split into `planes` bf16 planes for the fp32-on-bf16-MFMA
    kernels (y3_conv_split.hip); the 3->32 stem keeps its fp32 HWIO kernel."""
    w32, scale, shift = prepare_conv_params(w_var, bn_vars=bn_vars, bias_var=bias_var)
    k, _, cin, cout = w_var.shape
    if cin == 3:
        return w32, scale, shift
    key = w_var.op_name + '#split%d' % planes
    hit = _param_cache.get(key)
    if hit is not None and hit[0] == w_var.version:
        return hit[1], scale, shift
    ws = torch.empty(planes * k * k * cout * cin, dtype=torch.bfloat16, device=w_var.tensor.device)
    _lib.check(_lib.lib().y3_pack_conv_weights_split(fw.context(), fw.ptr(w_var.tensor), k, cin, cout, planes,
                                                     fw.ptr(ws)))
    _param_cache[key] = (w_var.version, ws)
    return ws, scale, shift


def wino_eligible(k, stride, cin, cout, c_up=0):
    """True for the convs the Winograd kernel takes (the rule lives in the library: y3_conv_wino_eligible)."""
    d = _lib.ConvDesc(1, 8, 8, cin, c_up, cout, k, stride, 1)
    return _lib.lib().y3_conv_wino_eligible(ctypes.byref(d)) == 1


def prepare_conv_params_wino(w_var, bn_vars=None, bias_var=None, stride=1):
    """As prepare_conv_params, with the Winograd-transformed kernel for the eligible layers (the others keep the
    direct kernel's packing)."""
    w32, scale, shift = prepare_conv_params(w_var, bn_vars=bn_vars, bias_var=bias_var)
    k, _, cin, cout = w_var.shape
    if not wino_eligible(k, stride, cin, cout):
        return w32, scale, shift
    key = w_var.op_name + '#wino'
    hit = _param_cache.get(key)
    if hit is not None and hit[0] == w_var.version:
        return hit[1], scale, shift
    wu = pack_wino(w_var.tensor)
    _param_cache[key] = (w_var.version, wu)
    return wu, scale, shift


def prepare_conv_alt_wino44(w_var, stride=1):
    """The F(4x4,3x3) packing of a layer y3_conv_wino44_candidate names (None for the others): y3_net_forward runs such a
    layer on that kernel when the launch is large enough (y3_net_set_layer_alt)."""
    k, _, cin, cout = w_var.shape
    if not wino44_candidate(k, stride, cin, cout):
        return None
    key = w_var.op_name + '#wino44'
    hit = _param_cache.get(key)
    if hit is not None and hit[0] == w_var.version:
        return hit[1]
    wu = pack_wino44(w_var.tensor)
    _param_cache[key] = (w_var.version, wu)
    return wu


def prepare_conv_params(w_var, bn_vars=None, bias_var=None):
    """Return (w_packed, scale, shift) device tensors for one conv layer.

    w_var: HWIO kernel variable.  bn_vars: (gamma, beta, moving_mean, moving_variance) or None.
    bias_var: bias variable or None.  Packing/folding run as HIP kernels through the C ABI and are cached
    until one of the variables is re-assigned.
    """
    vers = (w_var.version,) + tuple(v.version for v in (bn_vars or ())) + \
        ((bias_var.version,) if bias_var is not None else ())
    hit = _param_cache.get(w_var.op_name)
    if hit is not None and hit[0] == vers:
        return hit[1], hit[2], hit[3]
    L = _lib.lib()
    ctx = fw.context()
    k, _, cin, cout = w_var.shape
    w = w_var.tensor
    if cin == 3:
        w_dev = w  # the stem kernel consumes HWIO [27][32] directly
    else:
        w_dev = torch.empty(k * k * cout * cin, dtype=torch.float32, device=w.device)
        _lib.check(L.y3_pack_conv_weights(ctx, fw.ptr(w), k, cin, cout, fw.ptr(w_dev)))
    if bn_vars is not None:
        gamma, beta, mean, var = (v.tensor for v in bn_vars)
        scale = torch.empty(cout, dtype=torch.float32, device=w.device)
        shift = torch.empty(cout, dtype=torch.float32, device=w.device)
        _lib.check(L.y3_bn_fold(ctx, fw.ptr(gamma), fw.ptr(beta), fw.ptr(mean), fw.ptr(var),
                                ctypes.c_float(BN_EPS), cout, fw.ptr(scale), fw.ptr(shift)))
    else:
        scale = torch.ones(cout, dtype=torch.float32, device=w.device)
        shift = bias_var.tensor if bias_var is not None else torch.zeros(cout, dtype=torch.float32,
                                                                         device=w.device)
    _param_cache[w_var.op_name] = (vers, w_dev, scale, shift)
    return w_dev, scale, shift


_scratch = {}


def _conv_scratch(device, nbytes):
    """Per-(device, stream) scratch for the stream-K schedule (kernels on one stream are ordered, so one
    buffer per stream is enough)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _scratch.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _scratch[key] = t
    return t


def conv2d_fwd(x, w_dev, scale, shift, k, stride, cout, act, residual=None, x_up=None, use_workspace=True,
               planes=0, stats=None):
    """y = act(conv(x) * scale + shift) + residual on NHWC fp32 device tensors (y3_conv2d_fwd).
    use_workspace=False forces the data-parallel schedule (no scratch).  planes=2/3: w_dev is the split-plane
    packing and the products are rebuilt on the bf16 matrix pipe (y3_conv2d_fwd_split)."""
    n, h, w, cx = x.shape
    c_up = 0
    if x_up is not None:
        c_up = x_up.shape[3]
        if tuple(x_up.shape[:3]) != (n, h // 2, w // 2):
            raise ValueError("x_up must be [N, H/2, W/2, C] for the fused upsample+concat input")
    d = _lib.ConvDesc(n, h, w, cx + c_up, c_up, cout, k, stride, 1 if act else 0)
    y = torch.empty((n, h // stride, w // stride, cout), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws, ws_bytes = None, 0
    if use_workspace:
        ws_bytes = L.y3_conv_workspace_bytes(ctypes.byref(d))
        ws = _conv_scratch(x.device, ws_bytes) if ws_bytes else None
    if planes:
        _lib.check(L.y3_conv2d_fwd_split(fw.context(x.device), ctypes.byref(d), int(planes), fw.ptr(x), fw.ptr(x_up),
                                         fw.ptr(w_dev), fw.ptr(scale), fw.ptr(shift), fw.ptr(residual),
                                         fw.ptr(y), fw.ptr(ws), ctypes.c_size_t(ws_bytes)))
        return y
    if stats is not None:       # also: per row block of y, the column sums of y and y^2 (y3_conv_stats_blocks rows)
        if residual is not None or x_up is not None:
            raise ValueError("conv2d_fwd(stats=...) takes no residual and no fused upsample input")
        _lib.check(L.y3_conv2d_fwd_stats(fw.context(x.device), ctypes.byref(d), fw.ptr(x), fw.ptr(w_dev), fw.ptr(scale),
                                         fw.ptr(shift), fw.ptr(y), fw.ptr(stats), fw.ptr(ws), ctypes.c_size_t(ws_bytes)))
        return y
    _lib.check(L.y3_conv2d_fwd(fw.context(x.device), ctypes.byref(d), fw.ptr(x), fw.ptr(x_up),
                               fw.ptr(w_dev), fw.ptr(scale), fw.ptr(shift), fw.ptr(residual),
                               fw.ptr(y), fw.ptr(ws), ctypes.c_size_t(ws_bytes)))
    return y


def conv2d_fwd_wino(x, w_wino, scale, shift, cout, act, residual=None, use_workspace=True, stats=None):
    """3x3 stride-1 conv in its Winograd F(2x2,3x3) form (y3_conv2d_fwd_wino); w_wino from pack_wino.
    use_workspace=False forces the one-workgroup-per-block schedule (no stream-K scratch)."""
    n, h, w, cin = x.shape
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, 3, 1, 1 if act else 0)
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws, ws_bytes = None, 0
    if use_workspace:
        ws_bytes = L.y3_conv_wino_workspace_bytes(ctypes.byref(d))
        ws = _conv_scratch(x.device, ws_bytes) if ws_bytes else None
    if stats is not None:
        if residual is not None:
            raise ValueError("conv2d_fwd_wino(stats=...) takes no residual")
        _lib.check(L.y3_conv2d_fwd_wino_stats(fw.context(x.device), ctypes.byref(d), fw.ptr(x), fw.ptr(w_wino),
                                              fw.ptr(scale), fw.ptr(shift), fw.ptr(y), fw.ptr(stats), fw.ptr(ws),
                                              ctypes.c_size_t(ws_bytes)))
        return y
    _lib.check(L.y3_conv2d_fwd_wino(fw.context(x.device), ctypes.byref(d), fw.ptr(x), fw.ptr(w_wino), fw.ptr(scale),
                                    fw.ptr(shift), fw.ptr(residual), fw.ptr(y), fw.ptr(ws), ctypes.c_size_t(ws_bytes)))
    return y


def pack_wino(w_hwio):
    """HWIO [3,3,cin,cout] fp32 device tensor -> Winograd-transformed packing (y3_pack_conv_weights_wino)."""
    _, _, cin, cout = w_hwio.shape
    out = torch.empty(16 * cin * cout, dtype=torch.float32, device=w_hwio.device)
    _lib.check(_lib.lib().y3_pack_conv_weights_wino(fw.context(w_hwio.device), fw.ptr(w_hwio), cin, cout, fw.ptr(out)))
    return out


def wino44_eligible(k, stride, cin, cout, c_up=0):
    """True for the convs the F(4x4,3x3) inference kernel takes (y3_conv_wino44_eligible)."""
    d = _lib.ConvDesc(1, 8, 8, cin, c_up, cout, k, stride, 0)
    return _lib.lib().y3_conv_wino44_eligible(ctypes.byref(d)) == 1


def wino44_candidate(k, stride, cin, cout, c_up=0):
    """True for the conv shapes worth an F(4x4,3x3) packing beside the F(2x2,3x3) one (y3_conv_wino44_candidate)."""
    d = _lib.ConvDesc(1, 8, 8, cin, c_up, cout, k, stride, 0)
    return _lib.lib().y3_conv_wino44_candidate(ctypes.byref(d)) == 1


def wino44_preferred(n, h, w, k, stride, cin, cout, c_up=0):
    """True when y3_net_forward ('f32_wino' mode) runs this conv - input [n, h, w, cin] - on the F(4x4,3x3) kernel."""
    d = _lib.ConvDesc(n, h, w, cin, c_up, cout, k, stride, 0)
    return _lib.lib().y3_conv_wino44_preferred(ctypes.byref(d)) == 1


def pack_wino44(w_hwio):
    """HWIO [3,3,cin,cout] fp32 device tensor -> the F(4x4,3x3) packing (36*cin*cout floats, y3_pack_conv_weights_wino44)."""
    _, _, cin, cout = w_hwio.shape
    out = torch.empty(36 * cin * cout, dtype=torch.float32, device=w_hwio.device)
    _lib.check(_lib.lib().y3_pack_conv_weights_wino44(fw.context(w_hwio.device), fw.ptr(w_hwio), cin, cout, fw.ptr(out)))
    return out


def conv2d_fwd_wino44(x, w_wino44, scale, shift, cout, act, residual=None, use_workspace=True, stats=None):
    """3x3 stride-1 conv in its Winograd F(4x4,3x3) form (y3_conv2d_fwd_wino44); w_wino44 from pack_wino44.
    use_workspace=True: the two-kernel form (input transform written once into the workspace, then the batched GEMMs);
    False: the one-kernel form.  stats: a [y3_conv_stats_blocks(d, 2), 2, cout] tensor -> the training form
    y3_conv2d_fwd_wino44_stats (column sums of y and y^2 per 16-tile block; no residual)."""
    n, h, w, cin = x.shape
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, 3, 1, 1 if act else 0)
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws, ws_bytes = None, 0
    if use_workspace:
        ws_bytes = L.y3_conv_wino44_workspace_bytes(ctypes.byref(d))
        ws = _conv_scratch(x.device, ws_bytes) if ws_bytes else None
    if stats is not None:
        assert residual is None
        _lib.check(L.y3_conv2d_fwd_wino44_stats(fw.context(x.device), ctypes.byref(d), fw.ptr(x), fw.ptr(w_wino44),
                                                fw.ptr(scale), fw.ptr(shift), fw.ptr(y), fw.ptr(stats), fw.ptr(ws),
                                                ctypes.c_size_t(ws_bytes)))
        return y
    _lib.check(L.y3_conv2d_fwd_wino44(fw.context(x.device), ctypes.byref(d), fw.ptr(x), fw.ptr(w_wino44), fw.ptr(scale),
                                      fw.ptr(shift), fw.ptr(residual), fw.ptr(y), fw.ptr(ws), ctypes.c_size_t(ws_bytes)))
    return y


def conv2d_dgrad_wino44(dz, w_hwio, cin, accumulate_into=None, use_workspace=True):
    """Data gradient of a stride-1 3x3 conv in F(4x4,3x3) form (y3_pack_conv_weights_wino44_dgrad + y3_conv2d_dgrad_wino44):
    dz [n,h,w,cout] (cout % 32 == 0), w_hwio the forward kernel [3,3,cin,cout] (cin % 64 == 0) -> dx [n,h,w,cin]."""
    n, h, w, cout = dz.shape
    L, ctx = _lib.lib(), fw.context(dz.device)
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, 3, 1, 0)
    wk = torch.empty(36 * cin * cout, dtype=torch.float32, device=dz.device)
    _lib.check(L.y3_pack_conv_weights_wino44_dgrad(ctx, fw.ptr(w_hwio), cin, cout, fw.ptr(wk)))
    ones, zeros = torch.ones(cin, device=dz.device), torch.zeros(cin, device=dz.device)
    dx = accumulate_into if accumulate_into is not None else torch.empty((n, h, w, cin), dtype=torch.float32, device=dz.device)
    ws, ws_bytes = None, 0
    if use_workspace:
        g = _lib.ConvDesc(n, h, w, cout, 0, cin, 3, 1, 0)            # the gradient conv: [n,h,w,cout] -> [n,h,w,cin]
        ws_bytes = L.y3_conv_wino44_workspace_bytes(ctypes.byref(g))
        ws = _conv_scratch(dz.device, ws_bytes) if ws_bytes else None
    _lib.check(L.y3_conv2d_dgrad_wino44(ctx, ctypes.byref(d), fw.ptr(dz), cout, fw.ptr(wk), fw.ptr(ones), fw.ptr(zeros),
                                        1 if accumulate_into is not None else 0, fw.ptr(dx), fw.ptr(ws),
                                        ctypes.c_size_t(ws_bytes)))
    return dx


def upsample_nearest(x, out_h, out_w):
    n, h, w, c = x.shape
    y = torch.empty((n, out_h, out_w, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().y3_upsample_nearest(fw.context(x.device), fw.ptr(x), n, h, w, c, out_h, out_w,
                                              fw.ptr(y)))
    return y


def concat_channels(a, b):
    if a.shape[:3] != b.shape[:3]:
        raise ValueError("concat: spatial shapes differ: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    n, h, w, ca = a.shape
    cb = b.shape[3]
    y = torch.empty((n, h, w, ca + cb), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().y3_concat_channels(fw.context(a.device), fw.ptr(a), ca, fw.ptr(b), cb, n * h * w,
                                             fw.ptr(y)))
    return y


def add(a, b):
    if a.shape != b.shape:
        raise ValueError("add: shapes differ: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    y = torch.empty_like(a)
    _lib.check(_lib.lib().y3_add(fw.context(a.device), fw.ptr(a), fw.ptr(b), a.numel(), fw.ptr(y)))
    return y
