"""Random y3f_sample jobs that cover the geometry parse_sample can produce (and the corners it rarely does): shared by
tests/test_feed_plan.py (host run of the device functions) and tests/test_feed_gpu.py (the kernels)."""
import numpy as np


def random_image(rng, h, w):
    kind = rng.randint(0, 3)
    if kind == 0:
        return rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    if kind == 1:        # smooth: gradients + a little noise (resampling filters see structure)
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 255 // max(h + w - 2, 1))], -1)
        return np.clip(img + rng.randint(-8, 9, size=img.shape), 0, 255).astype(np.uint8)
    return np.full((h, w, 3), rng.randint(0, 256, size=3), np.uint8)     # flat colour (greys included: s == 0)


def random_case(rng, out_size=None, interp=None):
    """kwargs of feed_native.make_job / sample."""
    h1, w1 = int(rng.randint(8, 90)), int(rng.randint(8, 110))
    img1 = random_image(rng, h1, w1)
    img2, lam = None, 1.0
    if rng.uniform() < 0.4:
        img2 = random_image(rng, int(rng.randint(8, 90)), int(rng.randint(8, 110)))
        lam = float(rng.beta(1.5, 1.5))
    mh = max(h1, img2.shape[0]) if img2 is not None else h1
    mw = max(w1, img2.shape[1]) if img2 is not None else w1
    colour = None
    if rng.uniform() < 0.7:
        colour = (int(rng.randint(-32, 33)) if rng.uniform() < 0.5 else 0,
                  int(rng.randint(-18, 19)) if rng.uniform() < 0.5 else None,
                  float(rng.uniform(0.5, 1.5)) if rng.uniform() < 0.5 else None,
                  float(rng.uniform(0.5, 1.5)) if rng.uniform() < 0.5 else None)
    offset, canvas_w, canvas_h = (0, 0), mw, mh
    if rng.uniform() < 0.5:          # expansion: the image somewhere on a larger black canvas
        ratio = rng.uniform(1, 4)
        canvas_w, canvas_h = int(mw * ratio), int(mh * ratio)
        offset = (int(rng.randint(0, canvas_w - mw + 1)), int(rng.randint(0, canvas_h - mh + 1)))
    kind = rng.randint(0, 6)
    if kind == 0:                    # the whole canvas
        window = (0, 0, canvas_w, canvas_h)
    elif kind == 5:                  # a window that may miss the image altogether / hang over the canvas edge
        ww, wh = int(rng.randint(1, canvas_w + 1)), int(rng.randint(1, canvas_h + 1))
        window = (int(rng.randint(-ww, canvas_w)), int(rng.randint(-wh, canvas_h)), ww, wh)
    else:
        ww, wh = int(rng.randint(max(1, canvas_w // 4), canvas_w + 1)), int(rng.randint(max(1, canvas_h // 4), canvas_h + 1))
        window = (int(rng.randint(0, canvas_w - ww + 1)), int(rng.randint(0, canvas_h - wh + 1)), ww, wh)
    if out_size is None:
        out_size = (int(rng.choice([32, 48, 64])),) * 2
    ow, oh = out_size
    mode = rng.randint(0, 5)
    if mode == 0:                    # plain resize
        resized, pad = (ow, oh), (0, 0)
    elif mode == 1:                  # letterbox
        scale = min(ow / window[2], oh / window[3])
        rw, rh = max(1, int(window[2] * scale)), max(1, int(window[3] * scale))
        resized, pad = (rw, rh), ((ow - rw) // 2, (oh - rh) // 2)
    elif mode == 2:                  # only one axis changes
        if rng.uniform() < 0.5:
            resized = (min(window[2], ow), int(rng.randint(1, oh + 1)))
        else:
            resized = (int(rng.randint(1, ow + 1)), min(window[3], oh))
        pad = (int(rng.randint(0, ow - resized[0] + 1)), int(rng.randint(0, oh - resized[1] + 1)))
    elif mode == 3:                  # exact halving / no resize at all where they fit
        if rng.uniform() < 0.5 and window[2] >= 2 and window[3] >= 2:
            window = (window[0], window[1], min(window[2] // 2 * 2, 2 * ow), min(window[3] // 2 * 2, 2 * oh))
        if window[2] % 2 == 0 and window[3] % 2 == 0 and window[2] // 2 <= ow and window[3] // 2 <= oh and rng.uniform() < 0.5:
            resized = (window[2] // 2, window[3] // 2)
        elif window[2] <= ow and window[3] <= oh:
            resized = (window[2], window[3])
        else:
            resized = (ow, oh)
        pad = (int(rng.randint(0, ow - resized[0] + 1)), int(rng.randint(0, oh - resized[1] + 1)))
    else:                            # anything that fits
        resized = (int(rng.randint(1, ow + 1)), int(rng.randint(1, oh + 1)))
        pad = (int(rng.randint(0, ow - resized[0] + 1)), int(rng.randint(0, oh - resized[1] + 1)))
    return dict(img1=img1, img2=img2, lam=lam, colour=colour, offset=offset, window=window,
                interp=int(rng.randint(0, 5)) if interp is None else int(interp), resized=resized, out_size=(ow, oh), pad=pad,
                pad_value=int(rng.choice([128, 0, 255, 7])), flip_x=bool(rng.uniform() < 0.5))


def describe(case):
    c = dict(case)
    c['img1'] = case['img1'].shape
    c['img2'] = None if case['img2'] is None else case['img2'].shape
    return repr(c)
