# coding: utf-8
"""`cv2` as the reference's demo script and its drawing helper see it (ref: test_single_image.py:38-44,77-83,
utils/plot_utils.py:19-29, utils/data_aug.py:285): numpy + PIL, BGR channel order like OpenCV.  imshow / waitKey do
nothing - there is no display on a GPU node.  Only reachable through yolov3_tensorflow_amd.compat (see there)."""
import numpy as np

from yolov3_tensorflow_amd.utils import data_utils as _du

__version__ = '4.0.0-yolo355-compat'

IMREAD_COLOR = 1
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4
COLOR_BGR2RGB, COLOR_RGB2BGR = 4, 4
FONT_HERSHEY_SIMPLEX = 0
LINE_8, LINE_AA = 8, 16
FILLED = -1


def imread(filename, flags=IMREAD_COLOR):
    """HxWx3 uint8 in B,G,R order; None when the file cannot be read (OpenCV's convention)."""
    from PIL import Image
    try:
        with Image.open(filename) as im:
            rgb = np.asarray(im.convert('RGB'))
    except (IOError, OSError):
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def imwrite(filename, img, params=None):
    from PIL import Image
    arr = np.asarray(img)
    if arr.ndim == 3 and arr.shape[2] == 3:
        arr = arr[:, :, ::-1]
    Image.fromarray(np.ascontiguousarray(arr.astype(np.uint8))).save(filename)
    return True


def resize(src, dsize, dst=None, fx=None, fy=None, interpolation=INTER_LINEAR):
    """dsize = (width, height).  INTER_NEAREST and INTER_LINEAR restate OpenCV's arithmetic (data_utils); the other
    modes fall back to INTER_LINEAR."""
    w, h = int(dsize[0]), int(dsize[1])
    if interpolation == INTER_NEAREST:
        return _du.resize_nearest_cv2(src, w, h)
    return _du.resize_bilinear_cv2(src, w, h)


def cvtColor(src, code):
    if code != COLOR_BGR2RGB:
        raise NotImplementedError("cv2 shim: only the BGR<->RGB swap is implemented")
    return np.ascontiguousarray(np.asarray(src)[:, :, ::-1])


def _draw(img):
    from PIL import Image, ImageDraw
    canvas = Image.fromarray(np.ascontiguousarray(np.asarray(img)[:, :, ::-1]))
    return canvas, ImageDraw.Draw(canvas)


def _commit(img, canvas):
    img[...] = np.asarray(canvas)[:, :, ::-1]
    return img


def rectangle(img, pt1, pt2, color, thickness=1, lineType=LINE_8, shift=0):
    canvas, d = _draw(img)
    rgb = tuple(int(c) for c in tuple(color)[:3][::-1])
    xs, ys = sorted((int(pt1[0]), int(pt2[0]))), sorted((int(pt1[1]), int(pt2[1])))      # any two opposite corners
    box = [xs[0], ys[0], xs[1], ys[1]]
    if thickness is not None and thickness < 0:
        d.rectangle(box, fill=rgb)
    else:
        d.rectangle(box, outline=rgb, width=max(int(thickness or 1), 1))
    return _commit(img, canvas)


def getTextSize(text, fontFace, fontScale, thickness):
    """((width, height), baseline) of PIL's default font scaled like Hershey simplex (~22 px tall at scale 1)."""
    h = max(int(round(22 * float(fontScale))), 6)
    return (int(round(0.6 * h * len(text))), h), max(int(round(0.45 * h)), 1)


def putText(img, text, org, fontFace, fontScale, color, thickness=1, lineType=LINE_8, bottomLeftOrigin=False):
    from PIL import ImageFont
    canvas, d = _draw(img)
    (_, h), _ = getTextSize(text, fontFace, fontScale, thickness)
    try:
        font = ImageFont.load_default(size=h)
    except TypeError:                       # older Pillow: fixed-size bitmap font
        font = ImageFont.load_default()
    rgb = tuple(int(c) for c in tuple(color)[:3][::-1])
    d.text((int(org[0]), int(org[1]) - h), text, fill=rgb, font=font)
    return _commit(img, canvas)


def imshow(winname, mat):
    pass


def waitKey(delay=0):
    return -1


def destroyAllWindows():
    pass
