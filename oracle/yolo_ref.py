"""CPU restatement of the reference forward graph and decode (TEST INFRASTRUCTURE — see oracle/__init__.py).

forward: utils/layer_utils.py:9-87 + model.py:30-80, with torch-CPU F.conv2d as the convolution engine
(an implementation independent of the HIP kernels), BN applied UNFOLDED exactly as model.py:35-41 states
it, fp32 or fp64.  decode: model.py:82-190 op by op in numpy at the requested dtype.
Parity unpinned for both (TensorFlow-defined ops; no TF available) — see oracle/__init__.py.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5      # model.py:37
LEAKY_ALPHA = 0.1  # model.py:48


class _Builder(object):
    """Tracks slim's variable naming (Conv, Conv_1, ... per scope) while the graph functions run."""

    def __init__(self, params, prefix, dtype):
        self.params, self.prefix, self.dtype = params, prefix, dtype
        self.scope, self.count = None, 0
        self.trace = []      # (name, output tensor NCHW) per conv, for per-layer debugging

    def enter(self, scope):
        self.scope, self.count = scope, 0

    def next_name(self):
        n = 'Conv' if self.count == 0 else 'Conv_%d' % self.count
        self.count += 1
        return '%s/%s/%s' % (self.prefix, self.scope, n)

    def p(self, name):
        return torch.from_numpy(np.asarray(self.params[name])).to(self.dtype)


def _slim_conv(b, x, filters, k, stride, padding, bn=True, act=True):
    """slim.conv2d under the arg_scope of model.py:43-49.  x is NCHW."""
    name = b.next_name()
    w = b.p(name + '/weights')                      # HWIO
    assert w.shape[3] == filters and w.shape[0] == k, (name, tuple(w.shape), filters, k)
    w_oihw = w.permute(3, 2, 0, 1).contiguous()
    pad = (k // 2) if padding == 'SAME' else 0
    y = F.conv2d(x, w_oihw, None, stride=stride, padding=pad)
    if bn:
        g, be = b.p(name + '/BatchNorm/gamma'), b.p(name + '/BatchNorm/beta')
        mu, var = b.p(name + '/BatchNorm/moving_mean'), b.p(name + '/BatchNorm/moving_variance')
        # inference batch norm, unfolded: (x - mean) * gamma / sqrt(var + eps) + beta
        inv = g / torch.sqrt(var + BN_EPS)
        y = (y - mu.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1) + be.view(1, -1, 1, 1)
    else:
        y = y + b.p(name + '/biases').view(1, -1, 1, 1)
    if act:
        y = torch.where(y > 0, y, LEAKY_ALPHA * y)   # tf.nn.leaky_relu = max(alpha*x, x)
    b.trace.append((name, y))
    return y


def conv2d(b, inputs, filters, kernel_size, strides=1):
    # utils/layer_utils.py:9-22
    if strides > 1:
        pad_total = kernel_size - 1
        pad_beg = pad_total // 2
        pad_end = pad_total - pad_beg
        inputs = F.pad(inputs, (pad_beg, pad_end, pad_beg, pad_end))   # W then H for NCHW
    return _slim_conv(b, inputs, filters, kernel_size, strides, 'SAME' if strides == 1 else 'VALID')


def darknet53_body(b, inputs):
    # utils/layer_utils.py:24-68
    def res_block(inputs, filters):
        shortcut = inputs
        net = conv2d(b, inputs, filters * 1, 1)
        net = conv2d(b, net, filters * 2, 3)
        return net + shortcut

    net = conv2d(b, inputs, 32, 3, strides=1)
    net = conv2d(b, net, 64, 3, strides=2)
    net = res_block(net, 32)
    net = conv2d(b, net, 128, 3, strides=2)
    for _ in range(2):
        net = res_block(net, 64)
    net = conv2d(b, net, 256, 3, strides=2)
    for _ in range(8):
        net = res_block(net, 128)
    route_1 = net
    net = conv2d(b, net, 512, 3, strides=2)
    for _ in range(8):
        net = res_block(net, 256)
    route_2 = net
    net = conv2d(b, net, 1024, 3, strides=2)
    for _ in range(4):
        net = res_block(net, 512)
    route_3 = net
    return route_1, route_2, route_3


def yolo_block(b, inputs, filters):
    # utils/layer_utils.py:71-79
    net = conv2d(b, inputs, filters * 1, 1)
    net = conv2d(b, net, filters * 2, 3)
    net = conv2d(b, net, filters * 1, 1)
    net = conv2d(b, net, filters * 2, 3)
    net = conv2d(b, net, filters * 1, 1)
    route = net
    net = conv2d(b, net, filters * 2, 3)
    return route, net


def upsample_layer(inputs, out_hw):
    # utils/layer_utils.py:82-87: resize_nearest_neighbor, align_corners=False -> src = floor(dst*in/out)
    n, c, h, w = inputs.shape
    oh, ow = out_hw
    iy = torch.clamp((torch.arange(oh, dtype=torch.float32) * (float(h) / oh)).floor().long(), max=h - 1)
    ix = torch.clamp((torch.arange(ow, dtype=torch.float32) * (float(w) / ow)).floor().long(), max=w - 1)
    return inputs[:, :, iy][:, :, :, ix]


def forward(params, x_nhwc, class_num=80, dtype=torch.float32, prefix='yolov3', return_trace=False):
    """model.py:30-80.  params: name -> ndarray (TF variable names without ':0').  x: [N,H,W,3].
    Returns three NHWC numpy feature maps (and the per-conv trace if asked)."""
    b = _Builder(params, prefix, dtype)
    x = torch.from_numpy(np.asarray(x_nhwc)).to(dtype).permute(0, 3, 1, 2).contiguous()
    det = 3 * (5 + class_num)
    with torch.no_grad():
        b.enter('darknet53_body')
        route_1, route_2, route_3 = darknet53_body(b, x)
        b.enter('yolov3_head')
        inter1, net = yolo_block(b, route_3, 512)
        fm1 = _slim_conv(b, net, det, 1, 1, 'SAME', bn=False, act=False)
        inter1 = conv2d(b, inter1, 256, 1)
        inter1 = upsample_layer(inter1, route_2.shape[2:])
        concat1 = torch.cat([inter1, route_2], dim=1)
        inter2, net = yolo_block(b, concat1, 256)
        fm2 = _slim_conv(b, net, det, 1, 1, 'SAME', bn=False, act=False)
        inter2 = conv2d(b, inter2, 128, 1)
        inter2 = upsample_layer(inter2, route_1.shape[2:])
        concat2 = torch.cat([inter2, route_1], dim=1)
        _, fm3 = yolo_block(b, concat2, 128)
        fm3 = _slim_conv(b, fm3, det, 1, 1, 'SAME', bn=False, act=False)
    outs = tuple(f.permute(0, 2, 3, 1).contiguous().numpy() for f in (fm1, fm2, fm3))
    if return_trace:
        return outs, [(n, t.permute(0, 2, 3, 1).contiguous().numpy()) for n, t in b.trace]
    return outs


# ---------------------------------------------------------------------------------------------------
# decode (model.py:82-190)
# ---------------------------------------------------------------------------------------------------
def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def reorg_layer(feature_map, anchors, img_size, class_num, dtype=np.float32):
    """model.py:82-137.  anchors: [3,2] (w,h); img_size: [H,W].  Returns x_y_offset, boxes (cx,cy,w,h),
    conf_logits, prob_logits."""
    fm = np.asarray(feature_map).astype(dtype)
    grid_size = fm.shape[1:3]                                             # [gh, gw]
    ratio = (np.asarray(img_size, np.float64) / np.asarray(grid_size, np.float64)).astype(np.float32)
    ratio = ratio.astype(dtype)                                           # [ratio_h, ratio_w]
    rescaled_anchors = np.asarray([(dtype(a[0]) / ratio[1], dtype(a[1]) / ratio[0]) for a in anchors], dtype)
    fm = fm.reshape(-1, grid_size[0], grid_size[1], 3, 5 + class_num)
    box_centers, box_sizes = fm[..., 0:2], fm[..., 2:4]
    conf_logits, prob_logits = fm[..., 4:5], fm[..., 5:]
    box_centers = _sigmoid(box_centers)
    grid_x, grid_y = np.meshgrid(np.arange(grid_size[1], dtype=np.int32), np.arange(grid_size[0], dtype=np.int32))
    x_y_offset = np.concatenate([grid_x.reshape(-1, 1), grid_y.reshape(-1, 1)], axis=-1)
    x_y_offset = x_y_offset.reshape(grid_size[0], grid_size[1], 1, 2).astype(dtype)
    box_centers = box_centers + x_y_offset
    box_centers = box_centers * ratio[::-1]
    box_sizes = np.exp(box_sizes) * rescaled_anchors
    box_sizes = box_sizes * ratio[::-1]
    boxes = np.concatenate([box_centers, box_sizes], axis=-1).astype(dtype)
    return x_y_offset, boxes, conf_logits, prob_logits


def predict(feature_maps, anchors, img_size, class_num, dtype=np.float32):
    """model.py:140-190 -> boxes [N,B,4] (x_min,y_min,x_max,y_max), confs [N,B,1], probs [N,B,C]."""
    anchors = np.asarray(anchors, np.float32).reshape(9, 2)
    groups = [anchors[6:9], anchors[3:6], anchors[0:3]]
    boxes_l, confs_l, probs_l = [], [], []
    for fm, anc in zip(feature_maps, groups):
        x_y_offset, boxes, conf_logits, prob_logits = reorg_layer(fm, anc, img_size, class_num, dtype)
        g = x_y_offset.shape[:2]
        boxes_l.append(boxes.reshape(-1, g[0] * g[1] * 3, 4))
        confs_l.append(_sigmoid(conf_logits.reshape(-1, g[0] * g[1] * 3, 1)))
        probs_l.append(_sigmoid(prob_logits.reshape(-1, g[0] * g[1] * 3, class_num)))
    boxes = np.concatenate(boxes_l, axis=1)
    confs = np.concatenate(confs_l, axis=1)
    probs = np.concatenate(probs_l, axis=1)
    cx, cy, w, h = boxes[..., 0:1], boxes[..., 1:2], boxes[..., 2:3], boxes[..., 3:4]
    two = dtype(2)
    boxes = np.concatenate([cx - w / two, cy - h / two, cx + w / two, cy + h / two], axis=-1)
    return boxes.astype(dtype), confs.astype(dtype), probs.astype(dtype)


# ---------------------------------------------------------------------------------------------------
# variables: creation-ordered names/shapes, synthetic initialisation, darknet file format
# ---------------------------------------------------------------------------------------------------
def variable_specs(class_num=80, prefix='yolov3'):
    """[(name, shape)] in TF creation order, derived by tracing the graph functions above with a shape-only
    parameter source (independent of the product's layer table)."""
    specs = []

    class _Probe(dict):
        def __missing__(self, name):
            raise KeyError(name)

    # run the graph symbolically on a tiny input, creating zero params on demand
    class _LazyBuilder(_Builder):
        def __init__(self):
            _Builder.__init__(self, {}, prefix, torch.float32)
            self.pending = None

        def p(self, name):
            return self.params[name]

    b = _LazyBuilder()

    # monkeypatch-free approach: wrap _slim_conv via a local function that registers shapes first
    def register(name, cin, cout, k, bn):
        specs.append((name + '/weights', (k, k, cin, cout)))
        b.params[name + '/weights'] = torch.zeros(k, k, cin, cout)
        if bn:
            for s in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                specs.append((name + '/BatchNorm/' + s, (cout,)))
                b.params[name + '/BatchNorm/' + s] = torch.ones(cout)
        else:
            specs.append((name + '/biases', (cout,)))
            b.params[name + '/biases'] = torch.zeros(cout)

    orig_next = b.next_name
    state = {}

    def traced_slim_conv(bb, x, filters, k, stride, padding, bn=True, act=True):
        # peek the name the real function will use
        cnt = bb.count
        name = '%s/%s/%s' % (bb.prefix, bb.scope, 'Conv' if cnt == 0 else 'Conv_%d' % cnt)
        register(name, x.shape[1], filters, k, bn)
        return _real_slim_conv(bb, x, filters, k, stride, padding, bn, act)

    global _slim_conv
    _real_slim_conv = _slim_conv
    _slim_conv = traced_slim_conv
    try:
        x = torch.zeros(1, 3, 32, 32)
        det = 3 * (5 + class_num)
        with torch.no_grad():
            b.enter('darknet53_body')
            r1, r2, r3 = darknet53_body(b, x)
            b.enter('yolov3_head')
            inter1, net = yolo_block(b, r3, 512)
            _slim_conv(b, net, det, 1, 1, 'SAME', bn=False, act=False)
            inter1 = upsample_layer(conv2d(b, inter1, 256, 1), r2.shape[2:])
            inter2, net = yolo_block(b, torch.cat([inter1, r2], 1), 256)
            _slim_conv(b, net, det, 1, 1, 'SAME', bn=False, act=False)
            inter2 = upsample_layer(conv2d(b, inter2, 128, 1), r1.shape[2:])
            _, f3 = yolo_block(b, torch.cat([inter2, r1], 1), 128)
            _slim_conv(b, f3, det, 1, 1, 'SAME', bn=False, act=False)
    finally:
        _slim_conv = _real_slim_conv
    return specs


_BODY_NON_RESIDUAL_3X3 = (0, 1, 4, 9, 26, 43)   # stem + the five stride-2 convs


def _is_residual_branch(name):
    """True for the 3x3 conv that closes a res_block (utils/layer_utils.py:25-32): in darknet53_body the
    odd-positioned 3x3 convs, i.e. every 3x3 conv except the stem and the stride-2 ones."""
    parts = name.split('/')
    if parts[-4] != 'darknet53_body':
        return False
    conv = parts[-3]
    idx = 0 if conv == 'Conv' else int(conv.split('_')[1])
    # within the body, 1x1 convs open a res_block and the following conv closes it
    return idx not in _BODY_NON_RESIDUAL_3X3 and idx in _RESIDUAL_CLOSERS


def _residual_closers():
    out, idx = set(), 2
    for blocks in (1, 2, 8, 8, 4):
        for _ in range(blocks):
            out.add(idx + 1)   # idx = the 1x1, idx+1 = the 3x3 closing the block
            idx += 2
        idx += 1               # the stride-2 conv that follows the stage
    return out


_RESIDUAL_CLOSERS = _residual_closers()


def synthetic_params(class_num=80, seed=1, prefix='yolov3'):
    """SURVEY.md §8(d) C2 synthetic weights: He-normal kernels, BN gamma~U(.8,1.2), beta~N(0,.05),
    mean~N(0,.05), var~U(.8,1.2); detection biases: conf channel -4.0, others N(0,.1).
    Two dampings keep activations and logits O(1) (documented in DESIGN.md): residual-branch gamma x0.25,
    detection kernels x0.25."""
    rng = np.random.RandomState(seed)
    params = OrderedDict()
    for name, shape in variable_specs(class_num, prefix):
        leaf = name.split('/')[-1]
        if leaf == 'weights':
            k, _, cin, _ = shape
            v = rng.normal(0.0, np.sqrt(2.0 / (k * k * cin)), size=shape)
            if shape[3] == 3 * (5 + class_num):
                v *= 0.25   # detection convs: logits ~N(0,2): finite exp(t_w), a few hundred confident boxes
        elif leaf == 'gamma':
            v = rng.uniform(0.8, 1.2, size=shape)
            if _is_residual_branch(name):
                v *= 0.25   # keep activations O(1) through the 23 residual adds (else std grows ~2^11.5)
        elif leaf == 'beta':
            v = rng.normal(0.0, 0.05, size=shape)
        elif leaf == 'moving_mean':
            v = rng.normal(0.0, 0.05, size=shape)
        elif leaf == 'moving_variance':
            v = rng.uniform(0.8, 1.2, size=shape)
        elif leaf == 'biases':
            v = rng.normal(0.0, 0.1, size=shape)
            v.reshape(3, 5 + class_num)[:, 4] = -4.0
        else:
            raise AssertionError(name)
        params[name] = v.astype(np.float32)
    return params


def write_darknet(params, path, header=(0, 2, 0, 0, 0)):
    """5 x int32, then per conv [beta,gamma,mean,var | bias] + kernel as (Cout,Cin,kh,kw)."""
    names = list(params.keys())
    out = []
    i = 0
    while i < len(names):
        assert names[i].endswith('/weights'), names[i]
        base = names[i][:-len('/weights')]
        hwio = params[names[i]]
        if i + 1 < len(names) and names[i + 1].startswith(base + '/BatchNorm/'):
            for s in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
                out.append(params[base + '/BatchNorm/' + s].ravel())
            i += 5
        else:
            out.append(params[base + '/biases'].ravel())
            i += 2
        out.append(np.ascontiguousarray(np.transpose(hwio, (3, 2, 0, 1))).ravel())
    with open(path, 'wb') as f:
        np.asarray(header, np.int32).tofile(f)
        np.concatenate(out).astype(np.float32).tofile(f)


def read_darknet(path, class_num=80, prefix='yolov3'):
    """Independent reader of the same format -> OrderedDict name -> ndarray (HWIO kernels)."""
    with open(path, 'rb') as f:
        np.fromfile(f, np.int32, 5)
        data = np.fromfile(f, np.float32)
    specs = variable_specs(class_num, prefix)
    params = OrderedDict()
    pos = 0
    i = 0
    while i < len(specs):
        name, shape = specs[i]
        base = name[:-len('/weights')]
        k, _, cin, cout = shape
        if i + 1 < len(specs) and '/BatchNorm/' in specs[i + 1][0]:
            for s in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
                params[base + '/BatchNorm/' + s] = data[pos:pos + cout].copy(); pos += cout
            i += 5
        else:
            params[base + '/biases'] = data[pos:pos + cout].copy(); pos += cout
            i += 2
        cnt = k * k * cin * cout
        params[name] = np.transpose(data[pos:pos + cnt].reshape(cout, cin, k, k), (2, 3, 1, 0)).copy()
        pos += cnt
    assert pos == data.size, (pos, data.size)
    # restore creation order
    return OrderedDict((n, params[n]) for n, _ in specs)
