# coding: utf-8
"""Box drawing without OpenCV (reference utils/plot_utils.py): the same per-class colour table (Python's `random`
seeded with 2, three draws per class) and a PIL rendering of one labelled box."""
from __future__ import division, print_function

import random

import numpy as np


def get_color_table(class_num, seed=2):
    """{class index: [r, g, b]} — identical values to the reference for the same (class_num, seed)."""
    rnd = random.Random(seed)
    return {c: [rnd.randint(0, 255), rnd.randint(0, 255), rnd.randint(0, 255)] for c in range(class_num)}


def plot_one_box(img, coord, label=None, color=None, line_thickness=None):
    '''
    Draw one box (and its caption on a filled strip above it) on `img` IN PLACE.
    img: HxWx3 uint8 numpy array.
    coord: [x_min, y_min, x_max, y_max] in pixels.
    label: caption string or None.  color: [r, g, b] or None (random).  line_thickness: pixels or None
    (default: 0.2 % of the longer image side, at least 1).
    '''
    from PIL import Image, ImageDraw
    thick = line_thickness or max(int(round(0.002 * max(img.shape[0:2]))), 1)
    rgb = tuple(int(v) for v in (color or [random.randint(0, 255) for _ in range(3)]))
    x0, y0, x1, y1 = (int(v) for v in coord[:4])
    canvas = Image.fromarray(img)
    draw = ImageDraw.Draw(canvas)
    draw.rectangle([x0, y0, x1, y1], outline=rgb, width=thick)
    if label:
        left, top, right, bottom = draw.textbbox((0, 0), label)
        tw, th = right - left, bottom - top
        draw.rectangle([x0, y0 - th - 4, x0 + tw + 2, y0], fill=rgb)
        draw.text((x0 + 1, y0 - th - 3), label, fill=(0, 0, 0))
    img[...] = np.asarray(canvas)
    return img
