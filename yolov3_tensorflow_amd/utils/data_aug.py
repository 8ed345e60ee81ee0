# coding: utf-8
"""The training augmentations of the reference's utils/data_aug.py (mix_up, random_color_distort, random_expand,
random_crop_with_constraints, resize_with_bbox, random_flip), OpenCV-free: numpy + PIL on RGB uint8 images.

These feed the train step (the feeder); they are host-side plumbing, not the hot path, and they are
random by construction, so they are held to DISTRIBUTIONAL fidelity, not bit parity:
  * the box arithmetic (crop constraints, IoU, clipping, flips, expansion offsets, mix-up weights) follows the reference
    line by line (utils/data_aug.py:12-380);
  * every function draws its random numbers from explicit generators (`rng`: numpy RandomState, `prng`: random.Random)
    instead of the process-global ones, so a feeder worker thread is reproducible from (seed, epoch, sample index) -
    the reference's tf.data threads share the global generators and are not (utils/data_utils.py:190 says so);
  * the colour jitter works in HSV with OpenCV's units (H in [0,180), S and V in [0,255]); the conversion itself goes
    through PIL's C code (hue rescaled), the exact 8-bit OpenCV definition is kept as rgb_to_hsv_u8 / hsv_to_rgb_u8;
  * cv2.resize with the random interpolation 0..4: INTER_NEAREST and INTER_LINEAR are the exact restatements of
    utils/data_utils.py; INTER_CUBIC / INTER_AREA / INTER_LANCZOS4 map to PIL's BICUBIC / BOX / LANCZOS (no OpenCV here).
"""
from __future__ import division, print_function

import random as _random

import numpy as np


def _gens(rng, prng):
    return (rng if rng is not None else np.random), (prng if prng is not None else _random)


def mix_up(img1, img2, bbox1, bbox2, rng=None):
    '''
    reference utils/data_aug.py:12-39.
    return:
        mix_img: HWC format mix up image
        mix_bbox: [N, 5] shape mix up bbox, i.e. `x_min, y_min, x_max, y_mix, mixup_weight`.
    '''
    rng, _ = _gens(rng, None)
    height = max(img1.shape[0], img2.shape[0])
    width = max(img1.shape[1], img2.shape[1])
    mix_img = np.zeros(shape=(height, width, 3), dtype='float32')
    rand_num = rng.beta(1.5, 1.5)
    rand_num = max(0, min(1, rand_num))
    mix_img[:img1.shape[0], :img1.shape[1], :] = img1.astype('float32') * rand_num
    mix_img[:img2.shape[0], :img2.shape[1], :] += img2.astype('float32') * (1. - rand_num)
    mix_img = mix_img.astype('uint8')
    # the last element of the 2nd dimention is the mix up weight
    bbox1 = np.concatenate((bbox1, np.full(shape=(bbox1.shape[0], 1), fill_value=rand_num)), axis=-1)
    bbox2 = np.concatenate((bbox2, np.full(shape=(bbox2.shape[0], 1), fill_value=1. - rand_num)), axis=-1)
    mix_bbox = np.concatenate((bbox1, bbox2), axis=0)
    return mix_img, mix_bbox


def bbox_crop(bbox, crop_box=None, allow_outside_center=True):
    """Crop bounding boxes to a slice area (x_min, y_min, width, height); reference utils/data_aug.py:42-93."""
    bbox = bbox.copy()
    if crop_box is None:
        return bbox
    if not len(crop_box) == 4:
        raise ValueError("Invalid crop_box parameter, requires length 4, given {}".format(str(crop_box)))
    if sum([int(c is None) for c in crop_box]) == 4:
        return bbox
    l, t, w, h = crop_box
    left = l if l else 0
    top = t if t else 0
    right = left + (w if w else np.inf)
    bottom = top + (h if h else np.inf)
    crop_bbox = np.array((left, top, right, bottom))
    if allow_outside_center:
        mask = np.ones(bbox.shape[0], dtype=bool)
    else:
        centers = (bbox[:, :2] + bbox[:, 2:4]) / 2
        mask = np.logical_and(crop_bbox[:2] <= centers, centers < crop_bbox[2:]).all(axis=1)
    # transform borders
    bbox[:, :2] = np.maximum(bbox[:, :2], crop_bbox[:2])
    bbox[:, 2:4] = np.minimum(bbox[:, 2:4], crop_bbox[2:4])
    bbox[:, :2] -= crop_bbox[:2]
    bbox[:, 2:4] -= crop_bbox[:2]
    mask = np.logical_and(mask, (bbox[:, :2] < bbox[:, 2:4]).all(axis=1))
    return bbox[mask]


def bbox_iou(bbox_a, bbox_b, offset=0):
    """IoU of every pair of boxes of two sets ([N,4+], [M,4+]) -> [N,M]; reference utils/data_aug.py:95-125."""
    if bbox_a.shape[1] < 4 or bbox_b.shape[1] < 4:
        raise IndexError("Bounding boxes axis 1 must have at least length 4")
    tl = np.maximum(bbox_a[:, None, :2], bbox_b[:, :2])
    br = np.minimum(bbox_a[:, None, 2:4], bbox_b[:, 2:4])
    area_i = np.prod(br - tl + offset, axis=2) * (tl < br).all(axis=2)
    area_a = np.prod(bbox_a[:, 2:4] - bbox_a[:, :2] + offset, axis=1)
    area_b = np.prod(bbox_b[:, 2:4] - bbox_b[:, :2] + offset, axis=1)
    return area_i / (area_a[:, None] + area_b - area_i)


def _iou_min_max(rows, crop):
    """min and max over the boxes of bbox_iou(box, crop) for ONE crop box, in plain Python floats: the same float64
    arithmetic as bbox_iou on a [N,4] float32 array against an int crop, without ~10 numpy calls per trial (the crop loop
    runs up to 300 trials per image)."""
    lo, hi = float('inf'), -float('inf')
    cl, ct, cr, cb = crop
    area_b = float((cr - cl) * (cb - ct))
    for x0, y0, x1, y1 in rows:
        tlx, tly = max(x0, cl), max(y0, ct)
        brx, bry = min(x1, cr), min(y1, cb)
        area_i = (brx - tlx) * (bry - tly) * (1.0 if (tlx < brx and tly < bry) else 0.0)
        v = area_i / ((x1 - x0) * (y1 - y0) + area_b - area_i)
        lo, hi = min(lo, v), max(hi, v)
    return lo, hi


def random_crop_with_constraints(bbox, size, min_scale=0.3, max_scale=1, max_aspect_ratio=2, constraints=None,
                                 max_trial=50, rng=None, prng=None):
    """SSD-style random crop under IoU constraints (reference utils/data_aug.py:128-225).
    Returns (cropped boxes [M,4+], (x_offset, y_offset, new_width, new_height))."""
    rng, prng = _gens(rng, prng)
    if constraints is None:
        constraints = ((0.1, None), (0.3, None), (0.5, None), (0.7, None), (0.9, None), (None, 1))
    w, h = size
    candidates = [(0, 0, w, h)]
    rows = [tuple(float(v) for v in b[:4]) for b in bbox]
    for min_iou, max_iou in constraints:
        min_iou = -np.inf if min_iou is None else min_iou
        max_iou = np.inf if max_iou is None else max_iou
        for _ in range(max_trial):
            scale = prng.uniform(min_scale, max_scale)
            aspect_ratio = prng.uniform(max(1 / max_aspect_ratio, scale * scale), min(max_aspect_ratio, 1 / (scale * scale)))
            crop_h = int(h * scale / np.sqrt(aspect_ratio))
            crop_w = int(w * scale * np.sqrt(aspect_ratio))
            if h - crop_h < 1 or w - crop_w < 1:     # (random.randrange(0) raises in the reference; a degenerate
                continue                             # trial is skipped here)
            crop_t = prng.randrange(h - crop_h)
            crop_l = prng.randrange(w - crop_w)
            if len(bbox) == 0:
                return bbox, (crop_l, crop_t, crop_w, crop_h)
            iou_min, iou_max = _iou_min_max(rows, (crop_l, crop_t, crop_l + crop_w, crop_t + crop_h))
            if min_iou <= iou_min and iou_max <= max_iou:
                candidates.append((crop_l, crop_t, crop_w, crop_h))
                break
    # random select one
    while candidates:
        crop = candidates.pop(rng.randint(0, len(candidates)))
        new_bbox = bbox_crop(bbox, crop, allow_outside_center=False)
        if new_bbox.size < 1:
            continue
        return new_bbox, (crop[0], crop[1], crop[2], crop[3])
    return bbox, (0, 0, w, h)


def rgb_to_hsv_u8(img):
    """cv2.cvtColor(uint8 RGB -> HSV): H = hue / 2 in [0,180), S = 255 * (V - min) / V, V = max, rounded to uint8."""
    v = img.astype(np.float32)
    r, g, b = v[..., 0], v[..., 1], v[..., 2]
    vmax = v.max(axis=-1)
    vmin = v.min(axis=-1)
    diff = vmax - vmin
    s = np.where(vmax > 0, diff * 255.0 / np.maximum(vmax, 1e-12), 0.0)
    d = np.where(diff > 0, diff, 1.0)
    h = np.where(vmax == r, (g - b) / d, np.where(vmax == g, 2.0 + (b - r) / d, 4.0 + (r - g) / d)) * 60.0
    h = np.where(diff > 0, h, 0.0)
    h = np.where(h < 0, h + 360.0, h) * 0.5
    out = np.stack([np.rint(h) % 180, np.rint(s), vmax], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def hsv_to_rgb_u8(hsv):
    """cv2.cvtColor(uint8 HSV -> RGB), the inverse of rgb_to_hsv_u8 (H in [0,180))."""
    x = hsv.astype(np.float32)
    h, s, v = x[..., 0] * 2.0 / 60.0, x[..., 1] / 255.0, x[..., 2]
    i = np.floor(h).astype(np.int32) % 6
    f = h - np.floor(h)
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([r, g, b], axis=-1)), 0, 255).astype(np.uint8)


def random_color_distort(img, brightness_delta=32, hue_vari=18, sat_vari=0.5, val_vari=0.5, rng=None):
    '''
    randomly distort image color: brightness, then hue / saturation / value in one of two orders
    (reference utils/data_aug.py:228-271).  img: RGB uint8, HWC (the reference holds BGR: only the conversion differs).
    '''
    rng, _ = _gens(rng, None)

    def random_hue(img_hsv, p=0.5):
        if rng.uniform(0, 1) > p:
            hue_delta = rng.randint(-hue_vari, hue_vari)
            img_hsv[:, :, 0] = (img_hsv[:, :, 0] + hue_delta) % 180
        return img_hsv

    def random_saturation(img_hsv, p=0.5):
        if rng.uniform(0, 1) > p:
            img_hsv[:, :, 1] *= 1 + rng.uniform(-sat_vari, sat_vari)
        return img_hsv

    def random_value(img_hsv, p=0.5):
        if rng.uniform(0, 1) > p:
            img_hsv[:, :, 2] *= 1 + rng.uniform(-val_vari, val_vari)
        return img_hsv

    if rng.uniform(0, 1) > 0.5:        # brightness
        img = img.astype(np.float32) + int(rng.uniform(-brightness_delta, brightness_delta))
    img = np.clip(img, 0, 255).astype(np.uint8)
    # RGB <-> HSV through PIL's C conversion (hue on a 0..255 circle there: rescaled to OpenCV's 0..180 so that the jitter
    # amounts mean what they mean in the reference); rgb_to_hsv_u8 / hsv_to_rgb_u8 above are the exact 8-bit OpenCV
    # definition in numpy, 10x slower - half of a feeder worker's time per image when they were used here
    from PIL import Image
    img_hsv = np.asarray(Image.fromarray(img).convert('HSV')).astype(np.float32)
    img_hsv[:, :, 0] *= 180.0 / 255.0
    if rng.randint(0, 2):
        img_hsv = random_hue(random_saturation(random_value(img_hsv)))
    else:
        img_hsv = random_value(random_hue(random_saturation(img_hsv)))
    img_hsv = np.clip(img_hsv, 0, 255)
    img_hsv[:, :, 0] = np.minimum(img_hsv[:, :, 0] * (255.0 / 180.0), 255.0)
    return np.asarray(Image.fromarray(img_hsv.astype(np.uint8), 'HSV').convert('RGB'))


def _resize_any(img, new_width, new_height, interp):
    from .data_utils import resize_nearest_cv2, resize_bilinear_cv2
    if interp == 0:
        return resize_nearest_cv2(img, new_width, new_height)
    if interp == 1:
        return resize_bilinear_cv2(img, new_width, new_height)
    from PIL import Image
    mode = {2: Image.BICUBIC, 3: Image.BOX, 4: Image.LANCZOS}[int(interp)]
    return np.asarray(Image.fromarray(np.asarray(img, np.uint8)).resize((int(new_width), int(new_height)), mode))


def letterbox_resize(img, new_width, new_height, interp=0):
    '''
    Letterbox resize. keep the original aspect ratio in the resized image (reference utils/data_aug.py:274-293).
    '''
    ori_height, ori_width = img.shape[:2]
    resize_ratio = min(new_width / ori_width, new_height / ori_height)
    resize_w = int(resize_ratio * ori_width)
    resize_h = int(resize_ratio * ori_height)
    img = _resize_any(img, resize_w, resize_h, interp)
    image_padded = np.full((new_height, new_width, 3), 128, np.uint8)
    dw = int((new_width - resize_w) / 2)
    dh = int((new_height - resize_h) / 2)
    image_padded[dh: resize_h + dh, dw: resize_w + dw, :] = img
    return image_padded, resize_ratio, dw, dh


def resize_with_bbox(img, bbox, new_width, new_height, interp=0, letterbox=False):
    '''
    Resize the image and correct the bbox accordingly (reference utils/data_aug.py:296-320; any of the five cv2
    interpolation codes).  bbox: [N, >=4]; extra columns (the mix-up weight) are kept.
    '''
    bbox = np.array(bbox, np.float32)
    if bbox.ndim == 1:
        bbox = bbox.reshape(-1, 4)
    if letterbox:
        image_padded, resize_ratio, dw, dh = letterbox_resize(img, new_width, new_height, interp)
        bbox[:, [0, 2]] = bbox[:, [0, 2]] * resize_ratio + dw
        bbox[:, [1, 3]] = bbox[:, [1, 3]] * resize_ratio + dh
        return image_padded, bbox
    ori_height, ori_width = img.shape[:2]
    img = _resize_any(img, new_width, new_height, interp)
    bbox[:, [0, 2]] = bbox[:, [0, 2]] / ori_width * new_width
    bbox[:, [1, 3]] = bbox[:, [1, 3]] / ori_height * new_height
    return img, bbox


def random_flip(img, bbox, px=0, py=0, rng=None):
    '''
    Randomly flip the image and correct the bbox (reference utils/data_aug.py:323-346).
    px / py: the probability of a horizontal / vertical flip
    '''
    rng, _ = _gens(rng, None)
    height, width = img.shape[:2]
    if rng.uniform(0, 1) < px:
        img = img[:, ::-1]
        xmax = width - bbox[:, 0]
        xmin = width - bbox[:, 2]
        bbox[:, 0] = xmin
        bbox[:, 2] = xmax
    if rng.uniform(0, 1) < py:
        img = img[::-1]
        ymax = height - bbox[:, 1]
        ymin = height - bbox[:, 3]
        bbox[:, 1] = ymin
        bbox[:, 3] = ymax
    return np.ascontiguousarray(img), bbox


def random_expand(img, bbox, max_ratio=4, fill=0, keep_ratio=True, prng=None):
    '''
    Random expand original image with borders: place it on a larger canvas (reference utils/data_aug.py:349-380).
    '''
    _, prng = _gens(None, prng)
    h, w, c = img.shape
    ratio_x = prng.uniform(1, max_ratio)
    ratio_y = ratio_x if keep_ratio else prng.uniform(1, max_ratio)
    oh, ow = int(h * ratio_y), int(w * ratio_x)
    off_y = prng.randint(0, oh - h)
    off_x = prng.randint(0, ow - w)
    dst = np.full(shape=(oh, ow, c), fill_value=fill, dtype=img.dtype)
    dst[off_y:off_y + h, off_x:off_x + w, :] = img
    # correct bbox
    bbox[:, :2] += (off_x, off_y)
    bbox[:, 2:4] += (off_x, off_y)
    return dst, bbox
