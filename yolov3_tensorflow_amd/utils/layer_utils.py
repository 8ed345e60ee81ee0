# coding: utf-8
"""Mirror of the reference's utils/layer_utils.py — same names, signatures and composition — with every
op executed by a HIP kernel through the C ABI (eager, op by op).  `yolov3.forward` does not go through
these functions: it runs the same graph as one fused launch plan (y3_net_forward).  Both paths share the
conv kernels and must agree bit for bit (tests/test_forward_gpu.py).
"""
from __future__ import division, print_function

from .. import engine
from .. import framework as fw


def _conv_layer(inputs, filters, kernel_size, stride, use_bn=True, activation=True, residual=None):
    """slim.conv2d under the arg_scope of model.py:43-49 (BN + leaky, no bias) or, with use_bn=False /
    activation=False, the detection-head form of model.py:55-57 (bias, linear)."""
    inputs = fw.as_device_f32(inputs)
    cin = int(inputs.shape[3])
    scope = fw.current_scope_name()
    name = fw.unique_layer_name('Conv')
    base = (scope + '/' if scope else '') + name
    w = fw.get_variable(base + '/weights', (kernel_size, kernel_size, cin, filters), fw.xavier_uniform)
    if use_bn:
        # creation order of tf.layers.BatchNormalization: gamma, beta, moving_mean, moving_variance
        bn = (fw.get_variable(base + '/BatchNorm/gamma', (filters,), fw.ones),
              fw.get_variable(base + '/BatchNorm/beta', (filters,), fw.zeros),
              fw.get_variable(base + '/BatchNorm/moving_mean', (filters,), fw.zeros, trainable=False),
              fw.get_variable(base + '/BatchNorm/moving_variance', (filters,), fw.ones, trainable=False))
        params = engine.prepare_conv_params(w, bn_vars=bn)
    else:
        bias = fw.get_variable(base + '/biases', (filters,), fw.zeros)
        params = engine.prepare_conv_params(w, bias_var=bias)
    return engine.conv2d_fwd(inputs, params[0], params[1], params[2], kernel_size, stride, filters,
                             activation, residual=residual)


def conv2d(inputs, filters, kernel_size, strides=1):
    # reference utils/layer_utils.py:9-22: stride>1 -> explicit pad (k-1)//2 both sides then VALID;
    # the kernel folds that padding into its load predicate (input row = oy*stride + ky - k//2).
    return _conv_layer(inputs, filters, kernel_size, strides)


# (filters of the stride-2 conv opening the stage, number of residual blocks) — utils/layer_utils.py:38-66
_DARKNET53_STAGES = ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4))


def res_block(inputs, filters):
    """1x1(filters) -> 3x3(2*filters) -> + shortcut, added AFTER the activation (utils/layer_utils.py:25-32)."""
    net = conv2d(inputs, filters, 1)
    net = conv2d(net, filters * 2, 3)
    return engine.add(net, inputs)


def darknet53_body(inputs):
    """52 convs: stem 3x3(32), then five stages each = stride-2 3x3 conv + residual blocks
    (reference utils/layer_utils.py:24-68).  Returns route_1 (/8, 256ch), route_2 (/16, 512ch),
    route_3 (/32, 1024ch)."""
    net = conv2d(inputs, 32, 3, strides=1)
    stage_out = []
    for filters, blocks in _DARKNET53_STAGES:
        net = conv2d(net, filters, 3, strides=2)
        for _ in range(blocks):
            net = res_block(net, filters // 2)
        stage_out.append(net)
    return stage_out[2], stage_out[3], stage_out[4]


def yolo_block(inputs, filters):
    """Alternating 1x1(filters) / 3x3(2*filters), six convs; the fifth output is the route
    (reference utils/layer_utils.py:71-79).  Returns (route, net)."""
    net = inputs
    route = None
    for i in range(6):
        if i % 2 == 0:
            net = conv2d(net, filters, 1)
        else:
            net = conv2d(net, filters * 2, 3)
        if i == 4:
            route = net
    return route, net


def upsample_layer(inputs, out_shape):
    """Nearest-neighbour resize to out_shape[1:3] = (height, width) (reference utils/layer_utils.py:82-87)."""
    return engine.upsample_nearest(fw.as_device_f32(inputs), int(out_shape[1]), int(out_shape[2]))
