# coding: utf-8
"""Training augmentations with the reference's `utils.data_aug` interface (mix_up, random_color_distort, random_expand,
random_crop_with_constraints, resize_with_bbox, random_flip, letterbox_resize, bbox_crop, bbox_iou), OpenCV-free: numpy +
PIL on RGB uint8 images.

These feed the train step (the feeder); they are host-side plumbing, not the hot path.  What is held
fixed against the reference (utils/data_aug.py:12-380), and how:
  * the box arithmetic - crop windows, IoU, clipping, flips, canvas offsets, mix-up weights - gives the reference's numbers
    draw for draw: tests/test_feeder_cpu.py replays vectors recorded from the reference module under the same seeds
    (tests/golden/make_aug_golden.py).  That pins the ORDER and the arguments of every random draw below, which is why
    each function says which draws it makes;
  * every function takes explicit generators (`rng`: a numpy RandomState, `prng`: a random.Random) and falls back to the
    process-global ones only when given none, so a feeder worker is reproducible from (seed, epoch, sample index) - the
    reference's tf.data threads share the global generators and are not (utils/data_utils.py:190 says so);
  * the colour jitter works in HSV with OpenCV's units (H in [0,180), S and V in [0,255]); the conversion itself goes
    through PIL's C code (hue rescaled), the exact 8-bit OpenCV definition is kept as rgb_to_hsv_u8 / hsv_to_rgb_u8;
  * cv2.resize with the random interpolation 0..4: INTER_NEAREST and INTER_LINEAR are the exact restatements of
    utils/data_utils.py; INTER_CUBIC / INTER_AREA / INTER_LANCZOS4 map to PIL's BICUBIC / BOX / LANCZOS (no OpenCV here).
"""
from __future__ import division, print_function

import math
import random as _random

import numpy as np

_INF = float('inf')


def _gens(rng, prng):
    """(numpy-style generator, random.Random-style generator), each defaulting to the process-global module."""
    return (np.random if rng is None else rng), (_random if prng is None else prng)


def _with_column(boxes, value):
    """boxes [N, C] with one more column holding `value` (float64, as np.full makes it)."""
    return np.hstack([boxes, np.full((len(boxes), 1), value)])


def mix_up_boxes(bbox1, bbox2, rng=None):
    """The image-free half of mix_up: draws the weight (one draw: rng.beta(1.5, 1.5)) and labels the boxes with it.
    Returns (weight of the first image, boxes [N1+N2, 5])."""
    rng, _ = _gens(rng, None)
    lam = min(1, max(0, rng.beta(1.5, 1.5)))
    return lam, np.vstack([_with_column(bbox1, lam), _with_column(bbox2, 1. - lam)])


def blend(img1, img2, lam):
    """The pixel half of mix_up: lam * img1 + (1 - lam) * img2 in float32 on a black canvas that holds both, as uint8."""
    (h1, w1), (h2, w2) = img1.shape[:2], img2.shape[:2]
    canvas = np.zeros((max(h1, h2), max(w1, w2), 3), np.float32)
    canvas[:h1, :w1] = img1.astype(np.float32) * lam
    canvas[:h2, :w2] += img2.astype(np.float32) * (1. - lam)
    return canvas.astype(np.uint8)


def mix_up(img1, img2, bbox1, bbox2, rng=None):
    """Blend two images on a common top-left-anchored canvas with a Beta(1.5, 1.5) weight (reference
    utils/data_aug.py:12-39).  One draw: rng.beta.
    Returns (uint8 HWC image, boxes [N1+N2, 5] = x_min, y_min, x_max, y_max, weight of the image the box came from)."""
    lam, boxes = mix_up_boxes(bbox1, bbox2, rng)
    return blend(img1, img2, lam), boxes


def bbox_crop(bbox, crop_box=None, allow_outside_center=True):
    """Boxes [N, 4+] re-expressed inside the window crop_box = (x_min, y_min, width, height): clipped to it, shifted to
    its origin, and dropped when nothing is left (or, with allow_outside_center=False, when the centre lies outside).
    A None / 0 entry of crop_box means "unbounded on that side" (reference utils/data_aug.py:42-93)."""
    out = bbox.copy()
    if crop_box is None:
        return out
    if len(crop_box) != 4:
        raise ValueError("Invalid crop_box parameter, requires length 4, given {}".format(str(crop_box)))
    if all(c is None for c in crop_box):
        return out
    x0, y0 = crop_box[0] or 0, crop_box[1] or 0
    origin = np.array((x0, y0))
    far = np.array((x0 + (crop_box[2] or _INF), y0 + (crop_box[3] or _INF)))
    if allow_outside_center:
        keep = np.ones(len(out), dtype=bool)
    else:
        mid = (out[:, :2] + out[:, 2:4]) / 2
        keep = ((origin <= mid) & (mid < far)).all(axis=1)
    out[:, :2] = np.maximum(out[:, :2], origin)
    out[:, 2:4] = np.minimum(out[:, 2:4], far)
    out[:, :2] -= origin
    out[:, 2:4] -= origin
    keep &= (out[:, :2] < out[:, 2:4]).all(axis=1)
    return out[keep]


def bbox_iou(bbox_a, bbox_b, offset=0):
    """Pairwise IoU of two box sets ([N, 4+], [M, 4+]) -> [N, M]; `offset` 1 counts pixels inclusively (reference
    utils/data_aug.py:95-125)."""
    if min(bbox_a.shape[1], bbox_b.shape[1]) < 4:
        raise IndexError("Bounding boxes axis 1 must have at least length 4")

    def area(lo, hi):
        side = hi - lo + offset
        return side[..., 0] * side[..., 1]

    lo = np.maximum(bbox_a[:, None, :2], bbox_b[:, :2])
    hi = np.minimum(bbox_a[:, None, 2:4], bbox_b[:, 2:4])
    overlap = area(lo, hi) * (lo < hi).all(axis=2)
    return overlap / (area(bbox_a[:, :2], bbox_a[:, 2:4])[:, None] + area(bbox_b[:, :2], bbox_b[:, 2:4]) - overlap)


def _iou_min_max(rows, crop):
    """min and max over the boxes of bbox_iou(box, crop) for ONE crop box, in plain Python floats: the same float64
    arithmetic as bbox_iou on a [N,4] float32 array against an int crop, without ~10 numpy calls per trial (the crop loop
    runs up to 300 trials per image)."""
    lo, hi = _INF, -_INF
    cl, ct, cr, cb = crop
    area_b = float((cr - cl) * (cb - ct))
    for x0, y0, x1, y1 in rows:
        tlx, tly = max(x0, cl), max(y0, ct)
        brx, bry = min(x1, cr), min(y1, cb)
        area_i = (brx - tlx) * (bry - tly) * (1.0 if (tlx < brx and tly < bry) else 0.0)
        v = area_i / ((x1 - x0) * (y1 - y0) + area_b - area_i)
        lo, hi = min(lo, v), max(hi, v)
    return lo, hi


_DEFAULT_IOU_BANDS = ((0.1, None), (0.3, None), (0.5, None), (0.7, None), (0.9, None), (None, 1))


def _draw_window(prng, w, h, min_scale, max_scale, max_aspect_ratio):
    """One trial window (x, y, width, height) inside a w x h image, or None for a degenerate one.  Draws, in this order:
    scale ~ U(min_scale, max_scale); aspect ~ U(max(1/R, s^2), min(R, 1/s^2)); then y and x offsets by randrange (the
    offsets are not drawn for a degenerate trial - random.randrange(0) raises in the reference there)."""
    s = prng.uniform(min_scale, max_scale)
    root = math.sqrt(prng.uniform(max(1 / max_aspect_ratio, s * s), min(max_aspect_ratio, 1 / (s * s))))
    win_h, win_w = int(h * s / root), int(w * s * root)
    if h - win_h < 1 or w - win_w < 1:
        return None
    y = prng.randrange(h - win_h)
    x = prng.randrange(w - win_w)
    return x, y, win_w, win_h


def _candidate_windows(prng, rows, w, h, min_scale, max_scale, max_aspect_ratio, bands, max_trial):
    """The trial loop of random_crop_with_constraints: per (floor, ceil) band, up to max_trial windows from _draw_window
    until one whose IoU with EVERY box row lies in the band.  Returns (windows found, None), or (None, window) when there
    are no rows: nothing to constrain, the first proper window is the crop."""
    found = []
    for floor, ceil in bands:
        for _ in range(max_trial):
            win = _draw_window(prng, w, h, min_scale, max_scale, max_aspect_ratio)
            if win is None:
                continue
            if not rows:
                return None, win
            x, y, ww, wh = win
            least, most = _iou_min_max(rows, (x, y, x + ww, y + wh))
            if floor <= least and most <= ceil:
                found.append(win)
                break
    return found, None


def _candidate_windows_native(prng, rows, w, h, min_scale, max_scale, max_aspect_ratio, bands, max_trial):
    """_candidate_windows in liby3feed.so (y3f_crop_candidates), drawing from prng's own Mersenne Twister state: the same
    windows, and prng left exactly where the Python loop would leave it (tests/test_feed_native.py).  getstate -> search ->
    setstate is not atomic: like any use of one generator from several threads, it needs a generator per thread (the feeder
    gives every sample its own)."""
    import ctypes
    from .. import feed_native
    version, words, gauss = prng.getstate()
    state = np.array(words, dtype=np.uint32)
    boxes = np.array(rows, dtype=np.float64).reshape(-1, 4)
    limits = np.array(bands, dtype=np.float64).reshape(-1, 2)
    windows = np.zeros((max(1, len(limits)), 4), np.int32)
    count = ctypes.c_int32(0)
    feed_native.check(feed_native.lib().y3f_crop_candidates(
        state.ctypes.data, boxes.ctypes.data, len(boxes), int(w), int(h), float(min_scale), float(max_scale),
        float(max_aspect_ratio), limits.ctypes.data, len(limits), int(max_trial), windows.ctypes.data, ctypes.byref(count)))
    prng.setstate((version, tuple(state.tolist()), gauss))
    if count.value < 0:
        return None, tuple(int(v) for v in windows[0])
    return [tuple(int(v) for v in win) for win in windows[:count.value]], None


def random_crop_with_constraints(bbox, size, min_scale=0.3, max_scale=1, max_aspect_ratio=2, constraints=None,
                                 max_trial=50, rng=None, prng=None):
    """SSD-style random crop (reference utils/data_aug.py:128-225).  For each (min_iou, max_iou) band of `constraints`,
    up to max_trial windows are drawn (prng, see _draw_window) until one whose IoU with EVERY box lies in the band; the
    whole image is always a candidate too.  Candidates are then taken in random order (rng.randint) until one keeps at
    least one box centre.  size = (width, height).
    Returns (boxes [M, 4+] in the window's frame, (x_offset, y_offset, width, height))."""
    rng, prng = _gens(rng, prng)
    w, h = size
    rows = [tuple(float(v) for v in b[:4]) for b in bbox]
    bands = [(-_INF if lo is None else lo, _INF if hi is None else hi)
             for lo, hi in (_DEFAULT_IOU_BANDS if constraints is None else constraints)]
    from .. import feed_native
    search = _candidate_windows
    if feed_native.enabled() and hasattr(prng, 'getstate') and max(w, h) < 2 ** 31:      # (any random.Random, or `random`)
        search = _candidate_windows_native
    found, only = search(prng, rows, w, h, min_scale, max_scale, max_aspect_ratio, bands, max_trial)
    if found is None:
        return bbox, only
    pool = [(0, 0, w, h)] + found
    while pool:
        win = pool.pop(rng.randint(0, len(pool)))
        kept = bbox_crop(bbox, win, allow_outside_center=False)
        if kept.size:
            return kept, tuple(win)
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


def color_distort_draws(brightness_delta=32, hue_vari=18, sat_vari=0.5, val_vari=0.5, rng=None):
    """The draws of random_color_distort, all from rng, in this order: a coin and, on heads, an integer-truncated
    brightness shift U(-brightness_delta, brightness_delta); one randint(0, 2) choosing the channel order (1: value,
    saturation, hue; 0: saturation, hue, value); then per channel a coin and, on heads, its amount - hue: an integer
    rotation randint(-hue_vari, hue_vari) on the 180-degree circle; saturation / value: a gain 1 + U(-vari, vari).
    Returns (brightness shift, hue rotation or None, saturation gain or None, value gain or None)."""
    rng, _ = _gens(rng, None)
    shift = int(rng.uniform(-brightness_delta, brightness_delta)) if rng.uniform(0, 1) > 0.5 else 0
    amount = [None, None, None]           # hue, saturation, value
    for channel in ((2, 1, 0) if rng.randint(0, 2) else (1, 0, 2)):
        if not rng.uniform(0, 1) > 0.5:
            continue
        if channel == 0:
            amount[0] = int(rng.randint(-hue_vari, hue_vari))
        else:
            vari = (sat_vari, val_vari)[channel - 1]
            amount[channel] = float(1 + rng.uniform(-vari, vari))
    return shift, amount[0], amount[1], amount[2]


def apply_color_distort(img, draws):
    """The pixel half of random_color_distort for draws = (brightness shift, hue rotation, saturation gain, value gain)."""
    from PIL import Image
    shift, hue, sat, val = draws
    if shift:
        img = img.astype(np.float32) + shift
    img = np.clip(img, 0, 255).astype(np.uint8)
    # RGB <-> HSV through PIL's C conversion (hue on a 0..255 circle there: rescaled to OpenCV's 0..180 so that the jitter
    # amounts mean what they mean in the reference); rgb_to_hsv_u8 / hsv_to_rgb_u8 above are the exact 8-bit OpenCV
    # definition in numpy, 10x slower - half of a feeder worker's time per image when they were used here
    hsv = np.asarray(Image.fromarray(img).convert('HSV')).astype(np.float32)
    hsv[..., 0] *= 180.0 / 255.0
    if hue is not None:
        hsv[..., 0] = (hsv[..., 0] + hue) % 180
    if sat is not None:
        hsv[..., 1] *= sat
    if val is not None:
        hsv[..., 2] *= val
    hsv = np.clip(hsv, 0, 255)
    hsv[..., 0] = np.minimum(hsv[..., 0] * (255.0 / 180.0), 255.0)
    return np.asarray(Image.fromarray(hsv.astype(np.uint8), 'HSV').convert('RGB'))


def random_color_distort(img, brightness_delta=32, hue_vari=18, sat_vari=0.5, val_vari=0.5, rng=None):
    """Photometric jitter of an RGB uint8 HWC image: brightness, then hue / saturation / value (reference
    utils/data_aug.py:228-271, which holds BGR: only the conversion differs).  The draws: color_distort_draws."""
    return apply_color_distort(img, color_distort_draws(brightness_delta, hue_vari, sat_vari, val_vari, rng))


def _resize_any(img, new_width, new_height, interp):
    """cv2.resize with interpolation code 0..4 (see the module docstring); the same bytes from liby3feed.so when the native
    path is on, from the numpy restatements / Pillow otherwise."""
    from .. import feed_native
    if feed_native.enabled():
        return feed_native.resize(img, new_width, new_height, interp)
    from .data_utils import resize_nearest_cv2, resize_bilinear_cv2
    if interp == 0:
        return resize_nearest_cv2(img, new_width, new_height)
    if interp == 1:
        return resize_bilinear_cv2(img, new_width, new_height)
    from PIL import Image
    mode = {2: Image.BICUBIC, 3: Image.BOX, 4: Image.LANCZOS}[int(interp)]
    return np.asarray(Image.fromarray(np.asarray(img, np.uint8)).resize((int(new_width), int(new_height)), mode))


def letterbox_geometry(src_w, src_h, new_width, new_height):
    """(scale, fitted width, fitted height, x padding, y padding) of letterbox_resize for a src_w x src_h image."""
    scale = min(new_width / src_w, new_height / src_h)
    fit_w, fit_h = int(scale * src_w), int(scale * src_h)
    return scale, fit_w, fit_h, int((new_width - fit_w) / 2), int((new_height - fit_h) / 2)


def letterbox_resize(img, new_width, new_height, interp=0):
    """Aspect-preserving resize onto a grey (128) new_height x new_width canvas, centred (reference
    utils/data_aug.py:274-293).  Returns (canvas, scale, x padding, y padding) - what maps a box into the canvas."""
    src_h, src_w = img.shape[:2]
    scale, fit_w, fit_h, pad_x, pad_y = letterbox_geometry(src_w, src_h, new_width, new_height)
    canvas = np.full((new_height, new_width, 3), 128, np.uint8)
    canvas[pad_y:pad_y + fit_h, pad_x:pad_x + fit_w] = _resize_any(img, fit_w, fit_h, interp)
    return canvas, scale, pad_x, pad_y


def resize_boxes(bbox, src_w, src_h, new_width, new_height, letterbox=False):
    """The box half of resize_with_bbox: a float32 copy of bbox ([N, >=4], or a flat list of 4k numbers) mapped from a
    src_w x src_h image into the resized (or letterboxed) one."""
    boxes = np.array(bbox, np.float32)
    if boxes.ndim == 1:
        boxes = boxes.reshape(-1, 4)
    xs, ys = boxes[:, 0:3:2], boxes[:, 1:4:2]          # views: x_min/x_max and y_min/y_max
    if letterbox:
        scale, _, _, pad_x, pad_y = letterbox_geometry(src_w, src_h, new_width, new_height)
        xs *= scale
        xs += pad_x
        ys *= scale
        ys += pad_y
    else:
        xs /= src_w
        xs *= new_width
        ys /= src_h
        ys *= new_height
    return boxes


def resize_with_bbox(img, bbox, new_width, new_height, interp=0, letterbox=False):
    """Resize an image (plain stretch, or letterbox) and carry its boxes along (reference utils/data_aug.py:296-320; any
    of the five cv2 interpolation codes).  bbox: [N, >=4]; columns past the fourth (the mix-up weight) pass through."""
    src_h, src_w = img.shape[:2]
    boxes = resize_boxes(bbox, src_w, src_h, new_width, new_height, letterbox)
    if letterbox:
        return letterbox_resize(img, new_width, new_height, interp)[0], boxes
    return _resize_any(img, new_width, new_height, interp), boxes


def flip_boxes(bbox, width, height, px=0, py=0, rng=None):
    """The box half of random_flip: two draws, always both (rng.uniform(0, 1) for x, then y); the boxes (modified in
    place) of a width x height image are mirrored with it.  Returns (flipped in x?, flipped in y?)."""
    rng, _ = _gens(rng, None)
    in_x = rng.uniform(0, 1) < px
    if in_x:
        bbox[:, [0, 2]] = width - bbox[:, [2, 0]]
    in_y = rng.uniform(0, 1) < py
    if in_y:
        bbox[:, [1, 3]] = height - bbox[:, [3, 1]]
    return in_x, in_y


def random_flip(img, bbox, px=0, py=0, rng=None):
    """Mirror the image left-right with probability px and top-bottom with probability py, boxes (modified in place)
    mirrored with it (reference utils/data_aug.py:323-346).  The draws: flip_boxes."""
    in_x, in_y = flip_boxes(bbox, img.shape[1], img.shape[0], px, py, rng)
    if in_x:
        img = img[:, ::-1]
    if in_y:
        img = img[::-1]
    return np.ascontiguousarray(img), bbox


def expand_boxes(bbox, src_w, src_h, max_ratio=4, keep_ratio=True, prng=None):
    """The box half of random_expand.  Draws from prng: the x ratio U(1, max_ratio), the y ratio (only when keep_ratio is
    false), then the y and the x offset by randint (inclusive); the boxes (modified in place) move by the offset.
    Returns (canvas width, canvas height, x offset, y offset)."""
    _, prng = _gens(None, prng)
    grow_x = prng.uniform(1, max_ratio)
    grow_y = grow_x if keep_ratio else prng.uniform(1, max_ratio)
    big_h, big_w = int(src_h * grow_y), int(src_w * grow_x)
    at_y = prng.randint(0, big_h - src_h)
    at_x = prng.randint(0, big_w - src_w)
    shift = (at_x, at_y)
    bbox[:, :2] += shift
    bbox[:, 2:4] += shift
    return big_w, big_h, at_x, at_y


def random_expand(img, bbox, max_ratio=4, fill=0, keep_ratio=True, prng=None):
    """Place the image at a random position on a `fill`-coloured canvas up to max_ratio times larger, boxes (modified in
    place) shifted with it (reference utils/data_aug.py:349-380).  The draws: expand_boxes."""
    src_h, src_w, channels = img.shape
    big_w, big_h, at_x, at_y = expand_boxes(bbox, src_w, src_h, max_ratio, keep_ratio, prng)
    canvas = np.full((big_h, big_w, channels), fill, dtype=img.dtype)
    canvas[at_y:at_y + src_h, at_x:at_x + src_w] = img
    return canvas, bbox
