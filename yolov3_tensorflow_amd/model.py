# coding=utf-8
"""Mirror of the reference's model.py: class yolov3 with forward / reorg_layer / predict /
loss_layer / box_iou / compute_loss (reference model.py:12-365).  Same constructor, same method
signatures, same return arities; tensors are NHWC fp32 device tensors (torch owns the memory), and all
arithmetic runs in HIP kernels behind include/yolo355.h.
"""
from __future__ import division, print_function

import ctypes

import numpy as np
import torch

from . import _lib
from . import engine
from . import framework as fw
from .utils.layer_utils import conv2d, darknet53_body, yolo_block, upsample_layer, _conv_layer

N_BODY_CONVS = 52  # utils/layer_utils.py:24-68
# compute_dtype -> y3_net_set_dtype code.  'f32': exact fp32 MFMA; 'f32_bf16x6' / 'f32_bf16x3': fp32 tensors,
# every product rebuilt from 6 / 3 bf16 plane products with fp32 accumulation; 'bf16': bf16 storage; 'f32_wino': exact
# fp32 arithmetic with the Winograd F(2x2,3x3) kernel for the stride-1 3x3 convs (direct kernel elsewhere).
NET_DTYPES = {'f32': 0, 'bf16': 1, 'f32_bf16x6': 2, 'f32_bf16x3': 3, 'f32_wino': 4}


_SIDE_STREAMS = {}      # (device index, parts) -> the side streams of yolov3._forward_on_streams, shared by every model

class yolov3(object):

    def __init__(self, class_num, anchors, use_label_smooth=False, use_focal_loss=False,
                 batch_norm_decay=0.999, weight_decay=5e-4, use_static_shape=True):
        # reference model.py:14-28
        self.class_num = class_num
        self.anchors = anchors
        self.batch_norm_decay = batch_norm_decay
        self.use_label_smooth = use_label_smooth
        self.use_focal_loss = use_focal_loss
        self.weight_decay = weight_decay
        self.use_static_shape = use_static_shape
        self._nets = {}   # (ctx key, scope, dtype) -> dict(handle, version, keepalive, workspace)
        self.inference_streams = 1    # > 1: inference forwards split the batch over that many HIP streams (opt-in)
        self.img_size = None
        # 'f32' (the reference's precision), 'f32_bf16x6' / 'f32_bf16x3' (fp32 tensors, products on the bf16
        # matrix pipe, see NET_DTYPES) or 'bf16' (bf16 storage, fp32 accumulation; BASELINE configs[4]);
        # an attribute rather than a constructor argument so that the reference's signature is unchanged
        self.compute_dtype = 'f32'

    # ------------------------------------------------------------------------------------------
    # variables
    # ------------------------------------------------------------------------------------------
    def _layer_table(self, net_handle):
        L = _lib.lib()
        n = L.y3_net_num_layers(net_handle)
        out = []
        for i in range(n):
            k, s, cin, cout, bn = (ctypes.c_int() for _ in range(5))
            _lib.check(L.y3_net_layer_info(net_handle, i, ctypes.byref(k), ctypes.byref(s),
                                           ctypes.byref(cin), ctypes.byref(cout), ctypes.byref(bn)))
            out.append((k.value, s.value, cin.value, cout.value, bool(bn.value)))
        return out

    def _ensure_variables(self, scope, table):
        """Create (or fetch) the 366 variables in the reference's creation order:
        <scope>/darknet53_body/Conv[_k]/{weights,BatchNorm/...}, <scope>/yolov3_head/Conv[_k]/..."""
        layers = []
        for i, (k, s, cin, cout, bn) in enumerate(table):
            sub, j = ('darknet53_body', i) if i < N_BODY_CONVS else ('yolov3_head', i - N_BODY_CONVS)
            base = (scope + '/' if scope else '') + sub + '/' + ('Conv' if j == 0 else 'Conv_%d' % j)
            w = fw.get_variable(base + '/weights', (k, k, cin, cout), fw.xavier_uniform)
            if bn:
                bnv = (fw.get_variable(base + '/BatchNorm/gamma', (cout,), fw.ones),
                       fw.get_variable(base + '/BatchNorm/beta', (cout,), fw.zeros),
                       fw.get_variable(base + '/BatchNorm/moving_mean', (cout,), fw.zeros, trainable=False),
                       fw.get_variable(base + '/BatchNorm/moving_variance', (cout,), fw.ones,
                                       trainable=False))
                layers.append((w, bnv, None))
            else:
                layers.append((w, None, fw.get_variable(base + '/biases', (cout,), fw.zeros)))
        return layers

    def _get_net(self, device):
        if self.compute_dtype not in NET_DTYPES:
            raise ValueError("compute_dtype must be one of %s" % (sorted(NET_DTYPES),))
        bf16 = self.compute_dtype == 'bf16'
        planes = engine.SPLIT_PLANES.get(self.compute_dtype, 0)
        ctx = fw.context(device)
        scope = fw.current_scope_name()
        key = (ctx.value, scope, self.compute_dtype)
        ent = self._nets.get(key)
        L = _lib.lib()
        if ent is None:
            h = ctypes.c_void_p()
            _lib.check(L.y3_net_create(ctx, int(self.class_num), ctypes.byref(h)))
            if NET_DTYPES[self.compute_dtype]:
                _lib.check(L.y3_net_set_dtype(h, NET_DTYPES[self.compute_dtype]))
            table = self._layer_table(h)
            ent = dict(handle=h, table=table, version=-1, keep=None, ws=None, ws_bytes=0,
                       layers=self._ensure_variables(scope, table))
            self._nets[key] = ent
        if ent['version'] != fw.global_version():
            keep = []
            for i, (w, bnv, bias) in enumerate(ent['layers']):
                if self.compute_dtype == 'f32_wino':
                    wp, sc, sh = engine.prepare_conv_params_wino(w, bn_vars=bnv, bias_var=bias,
                                                                 stride=ent['table'][i][1])
                elif planes:
                    wp, sc, sh = engine.prepare_conv_params_split(w, bn_vars=bnv, bias_var=bias, planes=planes)
                else:
                    prep = engine.prepare_conv_params_bf16 if bf16 else engine.prepare_conv_params
                    wp, sc, sh = prep(w, bn_vars=bnv, bias_var=bias)
                _lib.check(L.y3_net_set_layer(ent['handle'], i, fw.ptr(wp), fw.ptr(sc), fw.ptr(sh)))
                alt = None
                if self.compute_dtype == 'f32_wino':       # the F(4x4,3x3) packing beside it, where the library wants one
                    alt = engine.prepare_conv_alt_wino44(w, stride=ent['table'][i][1])
                    _lib.check(L.y3_net_set_layer_alt(ent['handle'], i, fw.ptr(alt)))
                keep.append((wp, sc, sh, alt))
            ent['keep'] = keep   # the library holds raw pointers: keep the tensors alive
            ent['version'] = fw.global_version()
        return ent

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward(self, inputs, is_training=False, reuse=False):
        """reference model.py:30-80.  inputs: [N,H,W,3] fp32 (numpy or device tensor), H,W % 32 == 0.
        Returns (feature_map_1 [N,H/32,W/32,3*(5+C)], feature_map_2 (/16), feature_map_3 (/8))."""
        x = fw.as_device_f32(inputs)
        if x.dim() != 4 or x.shape[3] != 3:
            raise ValueError("inputs must be [N, H, W, 3], got %s" % (tuple(x.shape),))
        n, h, w, _ = x.shape
        # the input img_size, form: [height, weight]
        self.img_size = [int(h), int(w)]
        if is_training:
            from . import training
            return training.forward_train(self, x)
        det = 3 * (5 + self.class_num)
        fms = [torch.empty((n, h // s, w // s, det), dtype=torch.float32, device=x.device) for s in (32, 16, 8)]
        ns = int(getattr(self, 'inference_streams', 1) or 1)
        if ns > 1 and n % ns == 0 and n // ns >= 4 and not getattr(self, '_profiling_on', False):
            self._forward_on_streams(x, fms, ns)
        else:
            self._forward_chunk(x, fms)
        return fms[0], fms[1], fms[2]

    def choose_inference_streams(self, inputs, candidates=(1, 2), iters=3, rounds=4):
        """Measure the inference forward of `inputs` with each stream count and keep the fastest in self.inference_streams.
        Whether two streams overlap depends on how the HIP runtime maps streams to its hardware queues (GPU_MAX_HW_QUEUES, 4
        by default): with RCCL initialised in the process the same code measured 12.1 ms on two streams against 10.9 on one,
        and 10.4 with 2 or 8 hardware queues (profiles/r04_streams_ab.txt) - so a caller that cannot set the variable before
        HIP starts asks the hardware.
        The candidates are timed in alternation (`rounds` times `iters` synchronised forwards each, after one untimed forward
        per candidate that builds its nets and workspaces) and compared by their BEST round: a single pass of three forwards
        each, taken while the clocks were still settling, picked one stream in a driver run where two were 4 % faster."""
        import time
        x = fw.as_device_f32(inputs)
        candidates = list(dict.fromkeys(int(c) for c in candidates))
        for ns in candidates:
            self.inference_streams = ns
            self.forward(x, False)
        torch.cuda.synchronize(x.device)
        best_t = {ns: float('inf') for ns in candidates}
        for _ in range(max(1, int(rounds))):
            for ns in candidates:
                self.inference_streams = ns
                t0 = time.perf_counter()
                for _ in range(iters):
                    self.forward(x, False)
                torch.cuda.synchronize(x.device)
                best_t[ns] = min(best_t[ns], time.perf_counter() - t0)
        best = min(candidates, key=lambda ns: best_t[ns])
        self.inference_streams = best
        self.inference_streams_timing = {ns: best_t[ns] / iters * 1e3 for ns in candidates}      # ms per forward, best round
        return best

    def _forward_chunk(self, x, fms):
        """One y3_net_forward on torch's CURRENT stream (the net, its context and its workspace belong to that stream)."""
        n, h, w, _ = x.shape
        ent = self._get_net(x.device)
        L = _lib.lib()
        need = L.y3_net_workspace_bytes(ent['handle'], n, h, w)
        if need == 0:
            _lib.check(_lib.Y3_EINVAL)
        if ent['ws'] is None or ent['ws_bytes'] < need:
            ent['ws'] = torch.empty(need, dtype=torch.uint8, device=x.device)
            ent['ws_bytes'] = need
        _lib.check(L.y3_net_forward(ent['handle'], fw.ptr(x), n, h, w, fw.ptr(ent['ws']),
                                    ctypes.c_size_t(ent['ws_bytes']), fw.ptr(fms[0]), fw.ptr(fms[1]),
                                    fw.ptr(fms[2])))

    def _forward_on_streams(self, x, fms, ns):
        """Inference shards by image (eval-mode BN): the batch as `ns` equal parts, part 0 on the caller's stream, the others
        on side streams of this model, every part through its own net / context / workspace, all writing into slices of the
        same output tensors.  One part's kernel tails, partly filled last rounds of blocks and launch gaps are filled by
        the other's kernels (BASELINE north star: "independent per-GPU streams for inference"; measured in
        profiles/r04_wino44.txt 8).  Opt-in: model.inference_streams = 2; forwards with layer profiling on stay on one
        stream, so per-layer event times are those of undisturbed kernels."""
        dev = x.device
        main = torch.cuda.current_stream(dev)
        # one set of side streams per (device, parts) for the whole PROCESS, not per model: streams beyond the hardware queues
        # share them (GPU_MAX_HW_QUEUES), and a process that had built five models with a side stream each ran its train step's
        # second stream on an occupied queue (bench.py's default line: 96 ms per step instead of 79.6, profiles/r06_wgrad_stream_ab.txt)
        key = (dev.index, ns)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(ns - 1)]
        pool = _SIDE_STREAMS[key]
        c = x.shape[0] // ns
        self._get_net(dev)                  # (parameters are prepared / packed on the caller's stream, once for all parts)
        for side in pool:
            side.wait_stream(main)          # the input and the packed parameters are ready - taken BEFORE part 0 is enqueued,
                                            # or the side streams would wait for part 0's kernels too
        self._forward_chunk(x[:c], [f[:c] for f in fms])
        for i, side in enumerate(pool, start=1):
            with torch.cuda.stream(side):
                self._forward_chunk(x[i * c:(i + 1) * c], [f[i * c:(i + 1) * c] for f in fms])
            for t in [x] + fms:
                t.record_stream(side)
        for side in pool:
            main.wait_stream(side)

    def forward_composed(self, inputs):
        """The same graph built op by op exactly as reference model.py:50-78 composes it (unfused
        upsample/concat/add kernels).  Exists to cross-check the fused launch plan."""
        x = fw.as_device_f32(inputs)
        self.img_size = [int(x.shape[1]), int(x.shape[2])]
        det = 3 * (5 + self.class_num)
        with fw.variable_scope('darknet53_body'):
            route_1, route_2, route_3 = darknet53_body(x)

        with fw.variable_scope('yolov3_head'):
            inter1, net = yolo_block(route_3, 512)
            feature_map_1 = _conv_layer(net, det, 1, 1, use_bn=False, activation=False)

            inter1 = conv2d(inter1, 256, 1)
            inter1 = upsample_layer(inter1, list(route_2.shape))
            concat1 = engine.concat_channels(inter1, route_2)

            inter2, net = yolo_block(concat1, 256)
            feature_map_2 = _conv_layer(net, det, 1, 1, use_bn=False, activation=False)

            inter2 = conv2d(inter2, 128, 1)
            inter2 = upsample_layer(inter2, list(route_1.shape))
            concat2 = engine.concat_channels(inter2, route_1)

            _, feature_map_3 = yolo_block(concat2, 128)
            feature_map_3 = _conv_layer(feature_map_3, det, 1, 1, use_bn=False, activation=False)

        return feature_map_1, feature_map_2, feature_map_3

    def set_layer_profiling(self, enabled, device=None):
        """Turn per-layer hipEvent recording on/off for the fused plan on the current stream."""
        ent = self._get_net(device if device is not None else fw.default_device())
        _lib.check(_lib.lib().y3_net_set_profiling(ent['handle'], 1 if enabled else 0))
        self._profiling_on = bool(enabled)      # (profiled forwards run on ONE stream: see _forward_on_streams)

    def read_layer_ms(self, device=None):
        """Per-layer ms averaged over the forwards recorded since the last read (synchronises), plus the
        layer table [(k, stride, cin, cout, has_bn)].  Every layer is ONE kernel launch."""
        ent = self._get_net(device if device is not None else fw.default_device())
        nl = len(ent['table'])
        buf = (ctypes.c_float * nl)()
        _lib.check(_lib.lib().y3_net_get_layer_ms(ent['handle'], buf, nl))
        return np.array(buf[:], dtype=np.float64), ent['table']

    def layer_is_streamk(self, n, h, w, device=None):
        """Per layer: does the launch plan run it on the persistent stream-K schedule at this input shape."""
        ent = self._get_net(device if device is not None else fw.default_device())
        L = _lib.lib()
        return np.array([L.y3_net_layer_is_streamk(ent['handle'], i, n, h, w) for i in range(len(ent['table']))], bool)

    def layer_fused(self, n, h, w, device=None):
        """Per layer (y3_net_layer_fused): 0 = own launch, 1 = runs inside the next layer's launch (its output never reaches
        memory, its profiled time is 0), 2 = its launch also runs the layer before it."""
        ent = self._get_net(device if device is not None else fw.default_device())
        L = _lib.lib()
        return np.array([L.y3_net_layer_fused(ent['handle'], i, n, h, w) for i in range(len(ent['table']))], int)

    def layer_times_ms(self, inputs, iters=5):
        """Per-layer hipEvent timing of the fused plan (for profiles/ and DESIGN.md tables)."""
        x = fw.as_device_f32(inputs)
        self.forward(x)
        self.set_layer_profiling(True, x.device)
        for _ in range(iters):
            self.forward(x)
        ms, table = self.read_layer_ms(x.device)
        self.set_layer_profiling(False, x.device)
        return ms, table

    # ------------------------------------------------------------------------------------------
    # decode
    # ------------------------------------------------------------------------------------------
    def reorg_layer(self, feature_map, anchors):
        '''
        reference model.py:82-137.
        feature_map: a feature_map from [feature_map_1, feature_map_2, feature_map_3] returned
            from `forward` function
        anchors: shape: [3, 2]
        returns x_y_offset [g,g,1,2], boxes [N,g,g,3,4] (cx,cy,w,h), conf_logits [N,g,g,3,1],
                prob_logits [N,g,g,3,class_num]
        '''
        fm = fw.as_device_f32(feature_map)
        n, gh, gw, ch = fm.shape
        if ch != 3 * (5 + self.class_num):
            raise ValueError("feature map has %d channels, expected %d" % (ch, 3 * (5 + self.class_num)))
        if self.img_size is None:
            raise ValueError("reorg_layer needs img_size: call forward() first")
        anc = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(3, 2))
        boxes = torch.empty((n, gh, gw, 3, 4), dtype=torch.float32, device=fm.device)
        _lib.check(_lib.lib().y3_reorg_boxes(fw.context(fm.device), fw.ptr(fm), n, gh, gw, int(self.class_num),
                                             int(self.img_size[0]), int(self.img_size[1]),
                                             anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                             fw.ptr(boxes)))
        fm5 = fm.view(n, gh, gw, 3, 5 + self.class_num)
        conf_logits = fm5[..., 4:5]
        prob_logits = fm5[..., 5:]
        # meshgrid of integer cell coordinates, last dim (x, y) (model.py:108-115); index plumbing only
        gx = torch.arange(gw, device=fm.device, dtype=torch.float32).view(1, gw, 1, 1).expand(gh, gw, 1, 1)
        gy = torch.arange(gh, device=fm.device, dtype=torch.float32).view(gh, 1, 1, 1).expand(gh, gw, 1, 1)
        x_y_offset = torch.cat([gx, gy], dim=-1)
        return x_y_offset, boxes, conf_logits, prob_logits

    def predict(self, feature_maps, with_scores=False):
        '''
        reference model.py:140-190.
        Receive the returned feature_maps from `forward` function,
        the produce the output predictions at the test stage.
        returns boxes [N,B,4] (x_min,y_min,x_max,y_max), confs [N,B,1], probs [N,B,class_num]
        (with_scores=True additionally returns scores = confs*probs, test_single_image.py:55, fused
        into the same pass).
        '''
        fm1, fm2, fm3 = (fw.as_device_f32(f) for f in feature_maps)
        n, g1h, g1w, ch = fm1.shape
        if ch != 3 * (5 + self.class_num):
            raise ValueError("feature map has %d channels, expected %d" % (ch, 3 * (5 + self.class_num)))
        if self.img_size is None:
            raise ValueError("predict needs img_size: call forward() first")
        h, w = self.img_size
        for f, s in ((fm1, 32), (fm2, 16), (fm3, 8)):
            if tuple(f.shape) != (n, h // s, w // s, ch):
                raise ValueError("feature map shape %s does not match input size %dx%d / %d" %
                                 (tuple(f.shape), h, w, s))
        B = 3 * sum((h // s) * (w // s) for s in (32, 16, 8))
        C = int(self.class_num)
        dev = fm1.device
        boxes = torch.empty((n, B, 4), dtype=torch.float32, device=dev)
        confs = torch.empty((n, B, 1), dtype=torch.float32, device=dev)
        probs = torch.empty((n, B, C), dtype=torch.float32, device=dev)
        scores = torch.empty((n, B, C), dtype=torch.float32, device=dev) if with_scores else None
        anc = np.ascontiguousarray(np.asarray(self.anchors, np.float32).reshape(9, 2))
        _lib.check(_lib.lib().y3_decode(fw.context(dev), fw.ptr(fm1), fw.ptr(fm2), fw.ptr(fm3), n, h, w, C,
                                        anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), fw.ptr(boxes),
                                        fw.ptr(confs), fw.ptr(probs), fw.ptr(scores)))
        if with_scores:
            return boxes, confs, probs, scores
        return boxes, confs, probs

    def detect(self, inputs, max_boxes=200, score_thresh=0.3, nms_thresh=0.45):
        """SURVEY §8(f) row 2 — the whole inference path for a batch with everything resident on the device:
        forward -> predict (+ fused conf*prob) -> per-class NMS for all N images in one launch set.  Replaces
        the per-image `sess.run` round trips of eval.py:114-123 / utils/eval_utils.py:237-261.
        Returns N tuples (boxes [K,4], scores [K], labels [K] int32) of device tensors as a list-like LazyDetections: the
        per-image counts come back asynchronously and the host waits for them only when the result is first indexed or
        iterated, so the next batch can be enqueued before this one is read."""
        from .utils import nms_utils
        fms = self.forward(inputs, False)
        boxes, _, _, scores = self.predict(fms, with_scores=True)
        return nms_utils.gpu_nms_batched(boxes, scores, self.class_num, max_boxes, score_thresh, nms_thresh, lazy=True)

    # ------------------------------------------------------------------------------------------
    # loss (training path)
    # ------------------------------------------------------------------------------------------
    def loss_layer(self, feature_map_i, y_true, anchors):
        from . import training
        return training.loss_layer(self, feature_map_i, y_true, anchors)

    def box_iou(self, pred_boxes, valid_true_boxes):
        from . import training
        return training.box_iou(pred_boxes, valid_true_boxes)

    def compute_loss(self, y_pred, y_true):
        from . import training
        return training.compute_loss(self, y_pred, y_true)
