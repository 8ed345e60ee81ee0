# coding: utf-8
"""The slice of the reference's utils/data_utils.py / utils/data_aug.py that sits immediately either side of
the hot path: annotation-line parsing (`parse_line`), target assignment (`process_box`,
on the device, batched), the OpenCV-free resizes / letterbox used by the single-image and eval paths, and the per-image /
per-batch feeder functions `parse_data` / `get_batch_data` (augmentations: utils/data_aug.py; the threaded, prefetching,
device-resident form: yolov3_tensorflow_amd/feeder.py)."""
from __future__ import division, print_function

import ctypes

import numpy as np
import torch

from .. import _lib
from .. import framework as fw


def parse_line(line):
    '''
    Given a line from the training/test txt file, return parsed info (reference utils/data_utils.py:15-48).
    line format: line_index, img_path, img_width, img_height, [box_info_1 (5 number)], ...
    return:
        line_idx: int
        pic_path: string.
        boxes: shape [N, 4] float32, [x_min, y_min, x_max, y_max] per ground-truth object
        labels: shape [N] int64, class index.
        img_width: int.
        img_height: int
    '''
    if isinstance(line, bytes):
        line = line.decode()
    fields = line.strip().split(' ')
    assert len(fields) > 8, ('Annotation error! Please check your annotation file. Make sure there is at least one '
                             'target object in each image.')
    line_idx, pic_path = int(fields[0]), fields[1]
    img_width, img_height = int(fields[2]), int(fields[3])
    obj = fields[4:]
    assert len(obj) % 5 == 0, ('Annotation error! Please check your annotation file. Maybe partially missing some '
                               'coordinates?')
    table = np.array(obj, dtype=object).reshape(-1, 5)
    labels = np.asarray([int(v) for v in table[:, 0]], np.int64)
    boxes = np.asarray([[float(v) for v in row] for row in table[:, 1:]], np.float32).reshape(-1, 4)
    return line_idx, pic_path, boxes, labels, img_width, img_height


def process_box(boxes, labels, img_size, class_num, anchors):
    '''
    Generate the y_true label, i.e. the ground truth feature_maps in 3 different scales
    (reference utils/data_utils.py:51-115), for ONE image, on the device.
    params:
        boxes: [N, 5] shape, float32 dtype. `x_min, y_min, x_max, y_mix, mixup_weight`.
        labels: [N] shape, int64 dtype.
        img_size: [width, height]
        class_num: int.
        anchors: [9, 2] shape, float32 dtype.
    returns y_true_13, y_true_26, y_true_52 as numpy arrays (like the reference).
    '''
    boxes = np.asarray(boxes, np.float32).reshape(1, -1, 5)
    labels = np.asarray(labels).reshape(1, -1)
    ys = process_box_batch(boxes, labels, [boxes.shape[1]], img_size, class_num, anchors)
    return tuple(y[0].cpu().numpy() for y in ys)


def process_box_batch(boxes, labels, counts, img_size, class_num, anchors):
    """Batched device form: boxes [N,Kmax,5], labels [N,Kmax], counts [N] -> three device tensors
    [N,g,g,3,6+C] ready for yolov3.compute_loss (no host round trip per image)."""
    b = fw.as_device_f32(boxes)
    if b.dim() != 3 or b.shape[2] != 5:
        raise ValueError("boxes must be [N, Kmax, 5]")
    n, kmax, _ = b.shape
    dev = b.device
    def as_i32(v, shape):        # numpy / list / (device) tensor -> contiguous int32 tensor on `dev`
        t = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
        return t.to(device=dev, dtype=torch.int32).reshape(shape).contiguous()
    lab = as_i32(labels, (n, kmax))
    cnt = as_i32(counts, (n,))
    w, h = int(img_size[0]), int(img_size[1])
    C = int(class_num)
    anc = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(9, 2))
    ys = [torch.empty((n, h // s, w // s, 3, 6 + C), dtype=torch.float32, device=dev) for s in (32, 16, 8)]
    _lib.check(_lib.lib().y3_process_box(fw.context(dev), fw.ptr(b), fw.ptr(lab), fw.ptr(cnt), n, kmax, C, w, h,
                                         anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                         fw.ptr(ys[0]), fw.ptr(ys[1]), fw.ptr(ys[2])))
    return ys


def resize_nearest_cv2(img, new_width, new_height):
    """cv2.resize(img, (new_width, new_height), interpolation=cv2.INTER_NEAREST) restated: src = min(floor(dst *
    src_size / dst_size), src_size - 1) (no half-pixel centre)."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    sx = np.minimum(np.floor(np.arange(new_width) * (w / new_width)).astype(np.int64), w - 1)
    sy = np.minimum(np.floor(np.arange(new_height) * (h / new_height)).astype(np.int64), h - 1)
    return img[sy][:, sx]


_COEF_BITS = 11      # OpenCV's INTER_RESIZE_COEF_BITS: uint8 bilinear runs in 2^11 fixed point


def _linear_taps(src, dst):
    """Per destination index: left source index, and the two 11-bit fixed-point weights, as OpenCV's resize computes
    them for INTER_LINEAR: fx = (dx + 0.5) * scale - 0.5 in float32 (half-pixel centres, no antialiasing), clamped at both
    borders; weights cvRound(w * 2048) as int16 (round-half-even)."""
    scale = 1.0 / (dst / float(src))                    # OpenCV: scale_x = 1. / inv_scale_x
    f = ((np.arange(dst) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= src - 1
    s[hi], f[hi] = src - 1, 0.0
    a1 = np.rint(f.astype(np.float32) * np.float32(1 << _COEF_BITS)).astype(np.int32)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(1 << _COEF_BITS)).astype(np.int32)
    return s, np.minimum(s + 1, src - 1), a0, a1


def resize_bilinear_cv2(img, new_width, new_height):
    """cv2.resize(img, (new_width, new_height)) / interpolation=cv2.INTER_LINEAR for uint8 images, restated without
    OpenCV (the reference's eval / validation / plain-resize preprocessing: utils/data_utils.py:167,
    test_single_image.py:43).  Follows OpenCV's fixed-point path: horizontal pass with 11-bit weights into int32, vertical
    pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16), +2, >>2; an exact 2x2 downscale is OpenCV's INTER_AREA fast path
    ((a+b+c+d+2)>>2).  Half-pixel centres and NO antialiasing on downscale (PIL's BILINEAR antialiases: not the same
    function).  Parity unpinned: no OpenCV is installable here; the arithmetic is restated from OpenCV's resize.cpp."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise ValueError("resize_bilinear_cv2 expects a uint8 image")
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w = img.shape[:2]
    if (new_width, new_height) == (w, h):
        return (img[:, :, 0] if squeeze else img).copy()
    if w == 2 * new_width and h == 2 * new_height:
        v = img.astype(np.int32)
        out = ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        return out[:, :, 0] if squeeze else out
    x0, x1, ax0, ax1 = _linear_taps(w, new_width)
    y0, y1, ay0, ay1 = _linear_taps(h, new_height)
    v = img.astype(np.int32)
    rows = v[:, x0] * ax0[None, :, None] + v[:, x1] * ax1[None, :, None]          # [h, new_w, c] int32, scaled 2^11
    s0, s1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (s0 >> 4)) >> 16) + ((ay1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def _resize(img, new_width, new_height, interp):
    if interp not in (0, 1):
        raise ValueError("interp must be 0 (cv2.INTER_NEAREST) or 1 (cv2.INTER_LINEAR); the other OpenCV interpolations "
                         "are only used by the training augmentation (utils.data_aug)")
    from .. import feed_native
    img = np.asarray(img)
    if feed_native.enabled() and img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3:
        return feed_native.resize(img, new_width, new_height, interp)      # the same bytes, natively (liby3feed.so)
    return resize_nearest_cv2(img, new_width, new_height) if interp == 0 else resize_bilinear_cv2(img, new_width, new_height)


def letterbox_resize(img, new_width, new_height, interp=0):
    '''
    Letterbox resize keeping the aspect ratio (reference utils/data_aug.py:274-293), without OpenCV.
    img: HxWx3 uint8.  interp 0 = cv2.INTER_NEAREST (the demo call site, test_single_image.py:40), 1 =
    cv2.INTER_LINEAR (the eval / validation call sites, utils/data_utils.py:167).
    returns image_padded, resize_ratio, dw, dh
    '''
    img = np.asarray(img)
    ori_height, ori_width = img.shape[:2]
    resize_ratio = min(new_width / ori_width, new_height / ori_height)
    resize_w = int(resize_ratio * ori_width)
    resize_h = int(resize_ratio * ori_height)
    resized = _resize(img, resize_w, resize_h, interp)
    image_padded = np.full((new_height, new_width, 3), 128, np.uint8)
    dw = int((new_width - resize_w) / 2)
    dh = int((new_height - resize_h) / 2)
    image_padded[dh: resize_h + dh, dw: resize_w + dw, :] = resized
    return image_padded, resize_ratio, dw, dh


def resize_with_bbox(img, bbox, new_width, new_height, interp=0, letterbox=False):
    '''
    Resize the image and correct the bbox accordingly (reference utils/data_aug.py:296-320).
    '''
    # columns 0-3 are scaled, any further column (parse_data hands over [N,5]: the mix-up weight) is kept, as in the
    # reference; a flat list of 4*k numbers is accepted as k boxes
    bbox = np.array(bbox, np.float32)
    if bbox.ndim == 1:
        bbox = bbox.reshape(-1, 4)
    if bbox.ndim != 2 or bbox.shape[1] < 4:
        raise ValueError("resize_with_bbox: bbox must be [N, >=4], got %s" % (bbox.shape,))
    if letterbox:
        image_padded, resize_ratio, dw, dh = letterbox_resize(img, new_width, new_height, interp)
        bbox[:, [0, 2]] = bbox[:, [0, 2]] * resize_ratio + dw
        bbox[:, [1, 3]] = bbox[:, [1, 3]] * resize_ratio + dh
        return image_padded, bbox
    ori_height, ori_width = np.asarray(img).shape[:2]
    img = _resize(img, new_width, new_height, interp)
    bbox[:, [0, 2]] = bbox[:, [0, 2]] / ori_width * new_width
    bbox[:, [1, 3]] = bbox[:, [1, 3]] / ori_height * new_height
    return img, bbox


# ---------------------------------------------------------------------------------------------------------------------
# the per-image / per-batch feeder functions (reference utils/data_utils.py:118-224)
# ---------------------------------------------------------------------------------------------------------------------
iter_cnt = 0        # the reference's module-global batch counter (multi-scale: a new size every `interval` batches)


def _read_rgb(pic_path):
    from PIL import Image
    img = Image.open(pic_path)
    return np.asarray(img if img.mode == 'RGB' else img.convert('RGB'))      # (convert() copies even RGB -> RGB)


def fix_crop_labels():
    """Opt-in departure from the reference (train.py --fix_crop_labels sets Y3_FIX_CROP_LABELS=1; an environment variable
    so that the feeder's worker processes see it too): the 'train' chain carries every box's class through the
    constrained crop, so a box keeps ITS label when an earlier box is dropped.  Off (the default) the reference's pairing is
    reproduced: boxes are filtered, labels are not (utils/data_utils.py:150-153, 105; see collate())."""
    import os
    return os.environ.get('Y3_FIX_CROP_LABELS', '0') == '1'


def parse_sample(line, img_size, mode, letterbox_resize, rng=None, prng=None, as_uint8=False, out=None, defer=False):
    """The image half of the reference's parse_data (utils/data_utils.py:118-172): read (PIL, RGB), mix-up when `line` is
    a pair, the 'train' augmentation chain (colour distortion, expansion, constrained crop, resize with a random
    interpolation, horizontal flip) or the plain 'val' resize.  Returns (img_idx, float32 RGB image in [0,1] of shape
    [img_size[1], img_size[0], 3], boxes [K,5] with the mix-up weight in column 4, labels [>=K]: the crop may drop boxes
    and, like the reference, does not drop their labels - see collate(); fix_crop_labels() is the opt-in correction).  Target assignment (process_box) is NOT done here: the feeder runs it on the device for the whole batch (process_box_batch).

    Two executions of the same recipe, bit-identical (tests/test_feed_native.py): every draw and all box arithmetic come
    from utils.data_aug either way; the pixels go through liby3feed.so in one pass (feed_native.enabled(), the default), or
    through data_aug's numpy / Pillow functions one augmentation at a time (Y3_FEED_NATIVE=0).  `out`: a [h, w, 3] array
    (uint8 with as_uint8, else float32) to write the image into - the native path fills it directly.  `defer`: return the
    pixel JOB (feed_native.PixelJob) in place of the image - the feeder then runs a whole batch of them on the device
    (feed_device.DevicePixels: the same bytes)."""
    from . import data_aug
    from .. import feed_native
    rng = rng if rng is not None else np.random
    train = str(mode) == 'train'
    width, height = int(img_size[0]), int(img_size[1])
    partner, lam = None, 1.0
    if not isinstance(line, (list, tuple)):
        img_idx, pic_path, boxes, labels, _, _ = parse_line(line)
        img = _read_rgb(pic_path)
        # expand the 2nd dimension, mix up weight default to 1.
        boxes = np.concatenate((boxes, np.full(shape=(boxes.shape[0], 1), fill_value=1., dtype=np.float32)), axis=-1)
    else:
        # the mix up case
        _, pic_path1, boxes1, labels1, _, _ = parse_line(line[0])
        img_idx, pic_path2, boxes2, labels2, _, _ = parse_line(line[1])
        img, partner = _read_rgb(pic_path1), _read_rgb(pic_path2)
        lam, boxes = data_aug.mix_up_boxes(boxes1, boxes2, rng=rng)
        labels = np.concatenate((labels1, labels2))
    carry = train and fix_crop_labels()
    if carry:                                           # the class rides along as a sixth column (every box function of
        boxes = np.concatenate((boxes, np.asarray(labels, np.float32)[:, None]), axis=-1)     # data_aug keeps columns 4+)

    def split(bx):                                      # -> (boxes [K,5], labels)
        bx = np.asarray(bx, np.float32)
        if not carry:
            return bx, np.asarray(labels, np.int64)
        return np.ascontiguousarray(bx[:, :5]), bx[:, 5].astype(np.int64)

    if defer and not feed_native.enabled():
        raise ValueError("parse_sample(defer=True) needs the native pixel path (Y3_FEED_NATIVE=0 is set)")
    if not feed_native.enabled():
        if partner is not None:
            img = data_aug.blend(img, partner, lam)
        image, boxes = _augment_numpy(img, boxes, width, height, train, letterbox_resize, rng, prng)
        # the input of yolo_v3 should be in range 0~1 (as_uint8: the caller divides - the feeder's workers hand the 8-bit
        # image to the parent, a quarter of the bytes to pickle, and the parent divides straight into its pinned buffer)
        image = np.ascontiguousarray(image, np.uint8) if as_uint8 else np.asarray(image, np.float32) / 255.
        if out is not None:
            out[...] = image
            image = out
        return (img_idx, image) + split(boxes)

    # the same recipe, pixels last: the draws and the boxes first (in parse_data's order), then one native pass
    src_h = max(img.shape[0], partner.shape[0]) if partner is not None else img.shape[0]
    src_w = max(img.shape[1], partner.shape[1]) if partner is not None else img.shape[1]
    colour, offset, window, interp, flip = None, (0, 0), (0, 0, src_w, src_h), 1, False
    if train:
        colour = data_aug.color_distort_draws(rng=rng)
        canvas_w, canvas_h = src_w, src_h
        if rng.uniform(0, 1) > 0.5:                     # random expansion with prob 0.5
            canvas_w, canvas_h, at_x, at_y = data_aug.expand_boxes(boxes, src_w, src_h, 4, prng=prng)
            offset = (at_x, at_y)
        boxes, window = data_aug.random_crop_with_constraints(boxes, (canvas_w, canvas_h), rng=rng, prng=prng)
        interp = rng.randint(0, 5)                      # resize with random interpolation
    boxes = data_aug.resize_boxes(boxes, window[2], window[3], width, height, letterbox_resize)
    if train:
        flip, _ = data_aug.flip_boxes(boxes, width, height, px=0.5, rng=rng)
    resized, pad = (width, height), (0, 0)
    if letterbox_resize:
        _, fit_w, fit_h, pad_x, pad_y = data_aug.letterbox_geometry(window[2], window[3], width, height)
        resized, pad = (fit_w, fit_h), (pad_x, pad_y)
    if defer:
        image = feed_native.make_job(img, partner, lam, colour, offset, window, interp, resized, (width, height), pad, 128, flip)
    else:
        image = feed_native.sample(img, partner, lam, colour, offset, window, interp, resized, (width, height), pad, 128, flip,
                                   out=out, as_float=not as_uint8)
    return (img_idx, image) + split(boxes)


def _augment_numpy(img, boxes, width, height, train, letterbox_resize, rng, prng):
    """parse_data's chain on the numpy / Pillow functions of utils.data_aug, one augmentation at a time."""
    from . import data_aug
    if train:
        img = data_aug.random_color_distort(img, rng=rng)
        if rng.uniform(0, 1) > 0.5:                     # random expansion with prob 0.5
            img, boxes = data_aug.random_expand(img, boxes, 4, prng=prng)
        h, w, _ = img.shape                             # random cropping
        boxes, crop = data_aug.random_crop_with_constraints(boxes, (w, h), rng=rng, prng=prng)
        x0, y0, w, h = crop
        img = img[y0: y0 + h, x0: x0 + w]
        interp = rng.randint(0, 5)                      # resize with random interpolation
        img, boxes = data_aug.resize_with_bbox(img, boxes, width, height, interp=interp, letterbox=letterbox_resize)
        return data_aug.random_flip(img, boxes, px=0.5, rng=rng)
    return resize_with_bbox(img, boxes, width, height, interp=1, letterbox=letterbox_resize)


def parse_data(line, class_num, img_size, anchors, mode, letterbox_resize):
    '''
    reference utils/data_utils.py:118-176 (same signature and returns; images are RGB from PIL instead of cv2's BGR
    converted to RGB).
    param:
        line: a line from the training/test txt file (or a [line1, line2] pair: the mix up case)
        class_num: totol class nums.
        img_size: the size of image to be resized to. [width, height] format.
        anchors: anchors.
        mode: 'train' or 'val'. When set to 'train', data_augmentation will be applied.
        letterbox_resize: whether to use the letterbox resize, i.e., keep the original aspect ratio in the resized image.
    '''
    img_idx, img, boxes, labels = parse_sample(line, img_size, _str(mode), letterbox_resize)
    y_true_13, y_true_26, y_true_52 = process_box(boxes, labels, img_size, class_num, anchors)
    return img_idx, img, y_true_13, y_true_26, y_true_52


def _str(v):
    return v.decode() if isinstance(v, bytes) else str(v)


def mix_up_lines(batch_line, rng=None, prng=None):
    """The mix-up pairing of get_batch_data (utils/data_utils.py:202-211): each line is paired, with probability 0.5,
    with another line of the same batch."""
    import random as _random
    rng = rng if rng is not None else np.random
    prng = prng if prng is not None else _random
    batch_line = list(batch_line)
    mix_lines = []
    for idx, line in enumerate(batch_line):
        others = batch_line[:idx] + batch_line[idx + 1:]
        if rng.uniform(0, 1) < 0.5 and others:
            mix_lines.append([line, prng.sample(others, 1)[0]])
        else:
            mix_lines.append(line)
    return mix_lines


def multi_scale_size(count, interval=10):
    """The image size get_batch_data picks for batch number `count` when multi_scale is on
    (utils/data_utils.py:193-197): random.seed(count // interval); one of 320 .. 608 (range(10, 20) * 32)."""
    import random as _random
    sizes = [[x * 32, x * 32] for x in range(10, 20)]
    return _random.Random(count // interval).sample(sizes, 1)[0]


def get_batch_data(batch_line, class_num, img_size, anchors, mode, multi_scale=False, mix_up=False, letterbox_resize=True,
                   interval=10):
    '''
    generate a batch of imgs and labels (reference utils/data_utils.py:179-224: same signature and returns).
    Target assignment runs on the device for the whole batch and comes back as numpy (the reference's return type); the
    device-resident form without that round trip is yolov3_tensorflow_amd.feeder.Feeder.
    '''
    global iter_cnt
    mode = _str(mode)
    if multi_scale and mode == 'train':
        img_size = multi_scale_size(iter_cnt, interval)
    iter_cnt += 1
    batch_line = [l for l in (batch_line.tolist() if hasattr(batch_line, 'tolist') else list(batch_line))]
    if mix_up and mode == 'train':
        batch_line = mix_up_lines(batch_line)
    samples = _map_samples(batch_line, img_size, mode, letterbox_resize)
    ids, images, boxes, labels, counts = collate(samples)
    ys = process_box_batch(boxes, labels, counts, img_size, int(class_num), anchors)
    return (np.asarray(ids, np.int64), images) + tuple(y.cpu().numpy() for y in ys)


# How many parse_sample calls of ONE get_batch_data batch run at once: the `num_parallel_calls` of the reference's
# `dataset.map(py_func(get_batch_data), num_parallel_calls=args.num_threads)` (train.py:37-43), which the compat tf.data shim
# hands over.  1 (the default): in the calling thread, on the process-global generators, exactly like one py_func call of
# the reference.  More: worker threads (the pixel work releases the GIL); a 'train' batch then gives every line its own
# generators, seeded in line order from the global numpy generator - the samples no longer depend on which thread ran
# first, but they are not the ones the single-threaded call would draw.
BATCH_WORKERS = 1
_BATCH_POOLS = {}


def _map_samples(lines, img_size, mode, letterbox_resize):
    import random as _random
    workers = min(int(BATCH_WORKERS), len(lines))
    if workers <= 1:
        return [parse_sample(line, img_size, mode, letterbox_resize) for line in lines]
    from concurrent.futures import ThreadPoolExecutor
    pool = _BATCH_POOLS.get(workers)
    if pool is None:
        pool = _BATCH_POOLS[workers] = ThreadPoolExecutor(workers, thread_name_prefix='y3-batch')
    if str(mode) == 'train':
        seeds = [int(v) for v in np.random.randint(0, 2 ** 31 - 1, size=len(lines))]
        jobs = [pool.submit(parse_sample, line, img_size, mode, letterbox_resize, np.random.RandomState(seed),
                            _random.Random(seed)) for line, seed in zip(lines, seeds)]
    else:
        jobs = [pool.submit(parse_sample, line, img_size, mode, letterbox_resize) for line in lines]
    return [j.result() for j in jobs]


def collate(samples, out_images=None, with_images=True):
    """[(img_idx, image, boxes [K,5], labels [K]), ...] -> (ids, images [n,h,w,3], boxes [n,Kmax,5], labels [n,Kmax],
    counts [n]) padded to the largest box count (the layout y3_process_box takes).  with_images=False: the images are
    pixel jobs still to be run (parse_sample(defer=True)); `images` comes back as None."""
    n = len(samples)
    kmax = max(1, max(len(s[2]) for s in samples))
    h, w = samples[0][1].shape[:2]
    images = None
    if with_images:
        images = out_images if out_images is not None else np.empty((n, h, w, 3), np.float32)
    boxes = np.zeros((n, kmax, 5), np.float32)
    labels = np.zeros((n, kmax), np.int64)
    counts = np.zeros((n,), np.int64)
    ids = []
    for i, (idx, img, b, l) in enumerate(samples):
        if not with_images:
            pass
        elif img.dtype == np.uint8:      # img.astype(float32) / 255. of parse_data, written in place (no temporaries)
            np.true_divide(img, np.float32(255.), out=images[i], dtype=np.float32)
        elif img.ctypes.data != images[i].ctypes.data:      # (a feeder thread has written its slot of out_images already)
            images[i] = img
        # REFERENCE QUIRK kept (SURVEY B.10 "do not fix silently"): the constrained crop of the 'train' chain drops boxes
        # (bbox_crop filters them) but parse_data never filters `labels`, and process_box labels box i with labels[i] -
        # so after a crop that removed an earlier box the remaining boxes carry their predecessors' classes
        # (utils/data_utils.py:150-153,105).  The pairing box i <-> labels[i] over the first len(boxes) entries is that.
        k = len(b)
        boxes[i, :k], labels[i, :k], counts[i] = b, l[:k], k
        ids.append(idx)
    return ids, images, boxes, labels, counts
