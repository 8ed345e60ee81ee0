# coding: utf-8
"""`utils.plot_utils` as the reference's scripts import it (ref: utils/plot_utils.py:9-34), for images held in OpenCV's
B,G,R order: the package's colour table, and a labelled box drawn through the cv2 shim's primitives."""
import random

import cv2

from yolov3_tensorflow_amd.utils.plot_utils import get_color_table          # noqa: F401

_BLACK = [0, 0, 0]
_FONT = 0           # cv2.FONT_HERSHEY_SIMPLEX


def _caption(img, anchor, text, colour, stroke):
    """`text` in black on a `colour` strip whose bottom-left corner is `anchor` (the box's top-left corner)."""
    pen = max(stroke - 1, 1)
    size = float(stroke) / 3
    (wide, tall), _baseline = cv2.getTextSize(text, _FONT, fontScale=size, thickness=pen)
    x, y = anchor
    cv2.rectangle(img, anchor, (x + wide, y - tall - 3), colour, -1)
    cv2.putText(img, text, (x, y - 2), _FONT, size, _BLACK, thickness=pen, lineType=cv2.LINE_AA)


def plot_one_box(img, coord, label=None, color=None, line_thickness=None):
    """Draw the box coord = [x_min, y_min, x_max, y_max] on `img` in place; `label`, when given, goes on a filled strip
    above it.  No colour -> a random one; no thickness -> 0.2 % of the longer image side, at least one pixel."""
    if not line_thickness:
        line_thickness = max(int(round(0.002 * max(img.shape[:2]))), 1)
    if not color:
        color = [random.randint(0, 255), random.randint(0, 255), random.randint(0, 255)]
    x0, y0, x1, y1 = [int(v) for v in coord[:4]]
    cv2.rectangle(img, (x0, y0), (x1, y1), color, thickness=line_thickness)
    if label:
        _caption(img, (x0, y0), label, color, line_thickness)
