# coding: utf-8
"""utils.plot_utils of the reference (ref: utils/plot_utils.py:9-34) for images in OpenCV's B,G,R order: the same colour
table, boxes and captions drawn through the cv2 shim."""
import random

import cv2

from yolov3_tensorflow_amd.utils.plot_utils import get_color_table          # noqa: F401


def plot_one_box(img, coord, label=None, color=None, line_thickness=None):
    """Box `coord` = [x_min, y_min, x_max, y_max] on `img` in place, with `label` on a filled strip above it."""
    thick = line_thickness or max(int(round(0.002 * max(img.shape[0:2]))), 1)
    color = color or [random.randint(0, 255) for _ in range(3)]
    top_left = (int(coord[0]), int(coord[1]))
    cv2.rectangle(img, top_left, (int(coord[2]), int(coord[3])), color, thickness=thick)
    if label:
        font_thick = max(thick - 1, 1)
        scale = float(thick) / 3
        (tw, th), _ = cv2.getTextSize(label, 0, fontScale=scale, thickness=font_thick)
        cv2.rectangle(img, top_left, (top_left[0] + tw, top_left[1] - th - 3), color, -1)
        cv2.putText(img, label, (top_left[0], top_left[1] - 2), 0, scale, [0, 0, 0], thickness=font_thick,
                    lineType=cv2.LINE_AA)
