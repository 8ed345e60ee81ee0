"""CPU restatement of the reference's TRAIN step with torch autograd (TEST INFRASTRUCTURE — oracle/__init__.py).

  forward (is_training=True) : utils/layer_utils.py:9-87 + model.py:30-80 with batch-statistics BN
                               (model.py:35-41; all 72 BN layers, train.py:74)
  loss                       : model.py:192-365 (loss_layer, box_iou, compute_loss)
  objective                  : loss[0] + l2_loss, l2 = sum over the 75 conv kernels of wd*sum(w^2)/2
                               (model.py:49, train.py:78)
  gradients / clip / update  : train.py:105-115, utils/misc_utils.py:151-161 (TF1 optimizer definitions)
  BN moving statistics       : decay form of tf.layers.BatchNormalization, unbiased variance into the moving
                               variance (TF fused batch norm)
Every op here is TensorFlow-defined in the reference: parity unpinned (no TensorFlow available).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LEAKY = 0.1


class TrainGraph(object):
    """Holds the parameters as torch leaves and replays the reference graph in training mode."""

    def __init__(self, params, class_num=80, dtype=torch.float64, prefix='yolov3', masks=None):
        self.class_num, self.dtype, self.prefix = class_num, dtype, prefix
        # masks: optional {conv name: bool NCHW tensor} = the LeakyReLU branch (pre-activation > 0) to take per element
        # instead of this graph's own sign test.  LeakyReLU makes the gradient a discontinuous function of the forward
        # values: an element whose pre-activation changes sign between two implementations (|u| ~ 1e-6 in fp32) flips a
        # 1 / 0.1 factor, and a fraction p of flipped elements moves a gradient tensor by ~sqrt(p) relative — measured
        # here: the CPU fp32 graph is 4e-2 (worst tensor) from the fp64 one with its own branches, 2e-5 with a smooth
        # activation.  Handing both sides the SAME branches compares the same piecewise-linear function.
        self.masks = masks
        self.p = OrderedDict()
        for k, v in params.items():
            t = torch.tensor(np.asarray(v), dtype=dtype)
            leaf = k.split('/')[-1]
            t.requires_grad_(leaf in ('weights', 'gamma', 'beta', 'biases'))
            self.p[k] = t
        self.batch_stats = OrderedDict()   # conv name -> (mean, biased var, unbiased var)
        self.trace = OrderedDict()         # conv name -> activation (NCHW), for per-layer checks
        self.keep_trace = True             # False: large no-grad forwards (bs=64 @416 in fp64 would pin ~48 GB)

    # ---- graph --------------------------------------------------------------------------------------
    def _conv(self, x, filters, k, stride=1, bn=True, act=True):
        name = '%s/%s/%s' % (self.prefix, self._scope, 'Conv' if self._count == 0 else 'Conv_%d' % self._count)
        self._count += 1
        w = self.p[name + '/weights'].permute(3, 2, 0, 1)
        if stride > 1:
            pt = k - 1
            x = F.pad(x, (pt // 2, pt - pt // 2, pt // 2, pt - pt // 2))
            z = F.conv2d(x, w, stride=stride)
        else:
            z = F.conv2d(x, w, padding=k // 2)
        if bn:
            mean = z.mean(dim=(0, 2, 3))
            var = z.var(dim=(0, 2, 3), unbiased=False)
            n = z.numel() // z.shape[1]
            self.batch_stats[name] = (mean.detach(), var.detach(), var.detach() * (n / max(n - 1.0, 1.0)))
            g, b = self.p[name + '/BatchNorm/gamma'], self.p[name + '/BatchNorm/beta']
            z = (z - mean.view(1, -1, 1, 1)) * (g / torch.sqrt(var + BN_EPS)).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        else:
            z = z + self.p[name + '/biases'].view(1, -1, 1, 1)
        if act:
            pos = (z > 0) if self.masks is None or name not in self.masks else self.masks[name]
            z = torch.where(pos, z, LEAKY * z)
        if self.keep_trace:
            self.trace[name] = z
        return z

    def _res(self, x, f):
        return self._conv(self._conv(x, f, 1), 2 * f, 3) + x

    def _yolo_block(self, x, f):
        for i in range(5):
            x = self._conv(x, f if i % 2 == 0 else 2 * f, 1 if i % 2 == 0 else 3)
        return x, self._conv(x, 2 * f, 3)

    def forward(self, x_nhwc):
        x = torch.as_tensor(np.asarray(x_nhwc), dtype=self.dtype).permute(0, 3, 1, 2)
        self.img_size = [x.shape[2], x.shape[3]]
        det = 3 * (5 + self.class_num)
        self._scope, self._count = 'darknet53_body', 0
        net = self._conv(x, 32, 3)
        routes = []
        for f, blocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
            net = self._conv(net, f, 3, stride=2)
            for _ in range(blocks):
                net = self._res(net, f // 2)
            routes.append(net)
        r1, r2, r3 = routes[2], routes[3], routes[4]
        self._scope, self._count = 'yolov3_head', 0
        inter1, net = self._yolo_block(r3, 512)
        fm1 = self._conv(net, det, 1, bn=False, act=False)
        inter1 = self._conv(inter1, 256, 1)
        inter1 = inter1.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)     # nearest 2x
        inter2, net = self._yolo_block(torch.cat([inter1, r2], 1), 256)
        fm2 = self._conv(net, det, 1, bn=False, act=False)
        inter2 = self._conv(inter2, 128, 1)
        inter2 = inter2.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        _, f3 = self._yolo_block(torch.cat([inter2, r1], 1), 128)
        fm3 = self._conv(f3, det, 1, bn=False, act=False)
        return [f.permute(0, 2, 3, 1) for f in (fm1, fm2, fm3)]      # NHWC views (autograd-connected)

    # ---- loss (model.py:192-365) --------------------------------------------------------------------
    def reorg(self, fm, anchors):
        N, gh, gw, _ = fm.shape
        C = self.class_num
        ratio = torch.tensor([self.img_size[0] / gh, self.img_size[1] / gw], dtype=torch.float32).to(self.dtype)
        ra = torch.tensor([[a[0] / ratio[1].item(), a[1] / ratio[0].item()] for a in anchors], dtype=self.dtype)
        fm = fm.reshape(N, gh, gw, 3, 5 + C)
        xy = torch.sigmoid(fm[..., 0:2])
        gx = torch.arange(gw, dtype=self.dtype).view(1, gw, 1, 1).expand(gh, gw, 1, 1)
        gy = torch.arange(gh, dtype=self.dtype).view(gh, 1, 1, 1).expand(gh, gw, 1, 1)
        offset = torch.cat([gx, gy], -1)
        rr = torch.stack([ratio[1], ratio[0]])
        xy = (xy + offset) * rr
        wh = torch.exp(fm[..., 2:4]) * ra * rr
        return offset, torch.cat([xy, wh], -1), fm[..., 4:5], fm[..., 5:]

    @staticmethod
    def box_iou(pred_boxes, valid_true_boxes):
        pxy, pwh = pred_boxes[..., None, 0:2], pred_boxes[..., None, 2:4]
        txy, twh = valid_true_boxes[:, 0:2], valid_true_boxes[:, 2:4]
        mins = torch.maximum(pxy - pwh / 2., txy - twh / 2.)
        maxs = torch.minimum(pxy + pwh / 2., txy + twh / 2.)
        wh = torch.clamp(maxs - mins, min=0.)
        inter = wh[..., 0] * wh[..., 1]
        parea = pwh[..., 0] * pwh[..., 1]
        tarea = (twh[..., 0] * twh[..., 1]).unsqueeze(0)
        return inter / (parea + tarea - inter + 1e-10)

    def loss_layer(self, fm, y_true, anchors, use_label_smooth=False, use_focal_loss=False):
        N = fm.shape[0]
        gh, gw = fm.shape[1:3]
        y_true = torch.as_tensor(np.asarray(y_true), dtype=self.dtype)
        anchors_t = torch.tensor(np.asarray(anchors, np.float32), dtype=self.dtype)
        ratio = torch.tensor([self.img_size[0] / gh, self.img_size[1] / gw], dtype=torch.float32).to(self.dtype)
        rr = torch.stack([ratio[1], ratio[0]])
        offset, pred_boxes, conf_logits, prob_logits = self.reorg(fm, anchors)
        object_mask = y_true[..., 4:5]
        ignore = []
        for n in range(N):
            valid = y_true[n, ..., 0:4][object_mask[n, ..., 0] > 0.5]
            if valid.shape[0] == 0:
                ignore.append(torch.ones(gh, gw, 3, dtype=self.dtype))     # reduce_max over empty = -inf < 0.5
                continue
            iou = self.box_iou(pred_boxes[n].detach(), valid)
            ignore.append((iou.max(dim=-1).values < 0.5).to(self.dtype))
        ignore_mask = torch.stack(ignore).unsqueeze(-1)
        pred_xy, pred_wh = pred_boxes[..., 0:2], pred_boxes[..., 2:4]
        true_xy = y_true[..., 0:2] / rr - offset
        pred_xy = pred_xy / rr - offset
        true_twth = y_true[..., 2:4] / anchors_t
        pred_twth = pred_wh / anchors_t
        true_twth = torch.where(true_twth == 0, torch.ones_like(true_twth), true_twth)
        pred_twth = torch.where(pred_twth == 0, torch.ones_like(pred_twth), pred_twth)
        true_twth = torch.log(torch.clamp(true_twth, 1e-9, 1e9))
        pred_twth = torch.log(torch.clamp(pred_twth, 1e-9, 1e9))
        box_loss_scale = 2. - (y_true[..., 2:3] / float(self.img_size[1])) * (y_true[..., 3:4] / float(self.img_size[0]))
        mix_w = y_true[..., -1:]
        xy_loss = torch.sum((true_xy - pred_xy) ** 2 * object_mask * box_loss_scale * mix_w) / N
        wh_loss = torch.sum((true_twth - pred_twth) ** 2 * object_mask * box_loss_scale * mix_w) / N
        bce = F.binary_cross_entropy_with_logits
        conf_pos = object_mask * bce(conf_logits, object_mask, reduction='none')
        conf_neg = (1 - object_mask) * ignore_mask * bce(conf_logits, object_mask, reduction='none')
        conf_loss = conf_pos + conf_neg
        if use_focal_loss:
            conf_loss = conf_loss * 1.0 * torch.abs(object_mask - torch.sigmoid(conf_logits)) ** 2.0
        conf_loss = torch.sum(conf_loss * mix_w) / N
        target = y_true[..., 5:-1]
        if use_label_smooth:
            target = (1 - 0.01) * target + 0.01 * 1. / self.class_num
        class_loss = torch.sum(object_mask * bce(prob_logits, target, reduction='none') * mix_w) / N
        return xy_loss, wh_loss, conf_loss, class_loss

    def compute_loss(self, fms, y_trues, anchors, use_label_smooth=False, use_focal_loss=False):
        anchors = np.asarray(anchors, np.float32).reshape(9, 2)
        groups = [anchors[6:9], anchors[3:6], anchors[0:3]]
        parts = [self.loss_layer(f, y, a, use_label_smooth, use_focal_loss) for f, y, a in zip(fms, y_trues, groups)]
        xy, wh, conf, cls = (sum(p[i] for p in parts) for i in range(4))
        return [xy + wh + conf + cls, xy, wh, conf, cls]

    def l2_loss(self, weight_decay):
        return sum(weight_decay * 0.5 * (v ** 2).sum() for k, v in self.p.items() if k.endswith('/weights'))


def clip_by_norm(g, clip):
    n = torch.sqrt((g ** 2).sum())
    return g * clip / torch.maximum(n, torch.tensor(clip, dtype=g.dtype))


def apply_update(kind, w, g, slots, lr, step, momentum=0.9, decay=0.9, beta1=0.9, beta2=0.999, eps=None):
    """TF1 update rules.  slots: dict of tensors, mutated.  Returns the new weight."""
    if kind == 'sgd':
        return w - lr * g
    if kind == 'momentum':
        a = slots.setdefault('accum', torch.zeros_like(w))
        a.mul_(momentum).add_(g)
        return w - lr * a
    if kind == 'adam':
        eps = 1e-8 if eps is None else eps
        m = slots.setdefault('m', torch.zeros_like(w)); v = slots.setdefault('v', torch.zeros_like(w))
        m.mul_(beta1).add_((1 - beta1) * g); v.mul_(beta2).add_((1 - beta2) * g * g)
        lr_t = lr * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
        return w - lr_t * m / (torch.sqrt(v) + eps)
    if kind == 'rmsprop':
        eps = 1e-10 if eps is None else eps
        ms = slots.setdefault('ms', torch.ones_like(w)); mom = slots.setdefault('mom', torch.zeros_like(w))
        ms.mul_(decay).add_((1 - decay) * g * g)
        mom.mul_(momentum).add_(lr * g / torch.sqrt(ms + eps))
        return w - mom
    raise ValueError(kind)


def train_step(params, x, y_trues, anchors, class_num=80, optimizer='sgd', lr=1e-4, weight_decay=5e-4,
               bn_decay=0.99, clip=100.0, update_scopes=None, use_label_smooth=False, use_focal_loss=False,
               dtype=torch.float64, slots=None, step=1, masks=None):
    """One reference train step.  Returns dict(loss=[5 floats], l2, grads, new_params, batch_stats).
    masks: see TrainGraph (LeakyReLU branches imposed from outside)."""
    g = TrainGraph(params, class_num, dtype, masks=masks)
    fms = g.forward(x)
    loss = g.compute_loss(fms, y_trues, anchors, use_label_smooth, use_focal_loss)
    l2 = g.l2_loss(weight_decay)
    total = loss[0] + l2
    names = [k for k, v in g.p.items() if v.requires_grad and
             (update_scopes is None or any(k.startswith(s) for s in update_scopes))]
    grads = torch.autograd.grad(total, [g.p[k] for k in names], allow_unused=True)
    slots = {} if slots is None else slots
    new_params = OrderedDict((k, v.detach().clone()) for k, v in g.p.items())
    out_grads = OrderedDict()
    for k, gr in zip(names, grads):
        if gr is None:
            continue
        gc = clip_by_norm(gr, clip)
        out_grads[k] = gc.numpy()
        new_params[k] = apply_update(optimizer, g.p[k].detach(), gc, slots.setdefault(k, {}), lr, step)
    for name, (mean, var_b, var_u) in g.batch_stats.items():
        mm, mv = name + '/BatchNorm/moving_mean', name + '/BatchNorm/moving_variance'
        new_params[mm] = g.p[mm] * bn_decay + mean * (1 - bn_decay)
        new_params[mv] = g.p[mv] * bn_decay + var_u * (1 - bn_decay)
    return dict(loss=[float(v) for v in loss], l2=float(l2), grads=out_grads,
                new_params=OrderedDict((k, v.numpy()) for k, v in new_params.items()),
                feature_maps=[f.detach().numpy() for f in fms], graph=g, slots=slots)


def reapply(params, ref, optimizer, lr, step=1):
    """new_params of `ref` (a train_step result) under another optimizer, from the same clipped gradients (first step:
    empty slots)."""
    new = OrderedDict(ref['new_params'])
    for k, g in ref['grads'].items():
        w = torch.tensor(np.asarray(params[k]), dtype=torch.float64)
        new[k] = apply_update(optimizer, w, torch.tensor(g, dtype=torch.float64), {}, lr, step).numpy()
    return new


def process_box(boxes, labels, img_size, class_num, anchors):
    """Restatement of utils/data_utils.py:51-115 (pinned by tests/golden against the reference's own function).
    boxes [K,5] (x0,y0,x1,y1,mix_w); img_size [W,H]; returns y_true_13, _26, _52."""
    anchors = np.asarray(anchors, np.float32).reshape(9, 2)
    boxes = np.asarray(boxes, np.float32)
    centers = (boxes[:, 0:2] + boxes[:, 2:4]) / 2
    sizes = boxes[:, 2:4] - boxes[:, 0:2]
    ys = [np.zeros((img_size[1] // s, img_size[0] // s, 3, 6 + class_num), np.float32) for s in (32, 16, 8)]
    for y in ys:
        y[..., -1] = 1.
    bs = sizes[:, None, :]
    mins = np.maximum(-bs / 2, -anchors / 2)
    maxs = np.minimum(bs / 2, anchors / 2)
    whs = maxs - mins
    iou = (whs[..., 0] * whs[..., 1]) / (bs[..., 0] * bs[..., 1] + anchors[:, 0] * anchors[:, 1] -
                                         whs[..., 0] * whs[..., 1] + 1e-10)
    best = np.argmax(iou, axis=1)
    for i, idx in enumerate(best):
        group = 2 - idx // 3
        stride = {0: 8., 1: 16., 2: 32.}[idx // 3]
        x = int(np.floor(centers[i, 0] / stride)); y = int(np.floor(centers[i, 1] / stride))
        k = idx % 3
        ys[group][y, x, k, :2] = centers[i]
        ys[group][y, x, k, 2:4] = sizes[i]
        ys[group][y, x, k, 4] = 1.
        ys[group][y, x, k, 5 + labels[i]] = 1.
        ys[group][y, x, k, -1] = boxes[i, -1]
    return ys[0], ys[1], ys[2]


def synthetic_targets(seed, n, img_size, class_num, anchors, max_boxes=10):
    """SURVEY §8(d) C4 targets: per image K~U{1..10} boxes, class U{0..C-1}, w,h~U(10,300) clipped to the
    image, centres uniform with the box inside the image, mix weight 1 -> process_box -> stacked y_true."""
    rng = np.random.RandomState(seed)
    W, H = img_size
    out = [[], [], []]
    for _ in range(n):
        K = rng.randint(1, max_boxes + 1)
        w = np.minimum(rng.uniform(10, 300, K), W - 2.0); h = np.minimum(rng.uniform(10, 300, K), H - 2.0)
        cx = rng.uniform(w / 2, W - w / 2); cy = rng.uniform(h / 2, H - h / 2)
        boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, np.ones(K)], 1).astype(np.float32)
        labels = rng.randint(0, class_num, K)
        for o, y in zip(out, process_box(boxes, labels, [W, H], class_num, anchors)):
            o.append(y)
    return [np.stack(o) for o in out]
