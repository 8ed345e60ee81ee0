# coding: utf-8
"""`cv2` as the reference's demo scripts and its drawing helper see it (ref: test_single_image.py:38-44,77-83,
video_test.py:42-52,66,108-115, utils/plot_utils.py:19-29, utils/data_aug.py:285): numpy + PIL, BGR channel order like
OpenCV.  imshow / waitKey do nothing - there is no display on a GPU node.  VideoCapture / VideoWriter read and write what
this stack can code itself: Motion-JPEG AVI (utils.video_utils).  Only reachable through yolov3_tensorflow_amd.compat."""
import os
import sys

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
    """dsize = (width, height).  INTER_NEAREST and INTER_LINEAR restate OpenCV's arithmetic; for 3-channel uint8 images
    INTER_CUBIC / INTER_AREA / INTER_LANCZOS4 are Pillow's bicubic / box / Lanczos resampling (utils.data_aug), for other
    layouts they fall back to INTER_LINEAR."""
    w, h = int(dsize[0]), int(dsize[1])
    arr = np.asarray(src)
    if arr.dtype == np.uint8 and arr.ndim == 3 and arr.shape[2] == 3 and interpolation in (0, 1, 2, 3, 4):
        from yolov3_tensorflow_amd.utils import data_aug
        return data_aug._resize_any(arr, w, h, int(interpolation))
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


# ---- video (ref: video_test.py:42-52,66,108-115) ----------------------------------------------------------------------
CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT = 3, 4, 5, 7


class VideoCapture(object):
    """cv2.VideoCapture over utils.video_utils.open_video: Motion-JPEG / uncompressed AVI, multi-frame images, a directory of
    frames.  As with OpenCV, a file that cannot be opened gives an object whose isOpened() is False, get() 0 and read()
    (False, None); why is printed once to stderr."""

    def __init__(self, filename=None, apiPreference=None):
        from yolov3_tensorflow_amd.utils import video_utils
        self._reader = None
        if filename is not None:
            try:
                self._reader = video_utils.open_video(str(filename))
            except IOError as e:
                print('cv2.VideoCapture: %s' % e, file=sys.stderr)

    def isOpened(self):
        return self._reader is not None

    def get(self, prop):
        r = self._reader
        if r is None:
            return 0.0
        return float({CAP_PROP_FRAME_WIDTH: r.width, CAP_PROP_FRAME_HEIGHT: r.height, CAP_PROP_FPS: r.fps,
                      CAP_PROP_FRAME_COUNT: r.frame_count}.get(int(prop), 0))

    def read(self):
        frame = self._reader.read() if self._reader is not None else None
        if frame is None:
            return False, None
        return True, np.ascontiguousarray(frame[:, :, ::-1])

    def release(self):
        if self._reader is not None:
            self._reader.close()
            self._reader = None


def VideoWriter_fourcc(c1, c2, c3, c4):
    return int.from_bytes((c1 + c2 + c3 + c4).encode('latin1'), 'little')


class VideoWriter(object):
    """cv2.VideoWriter that always encodes Motion-JPEG in an AVI container (the codec this stack has).  Asked for another
    codec - video_test.py:51 asks for 'mp4v' into video_result.mp4 - it writes <stem>.avi next to the requested path and says so
    on stderr, rather than put an AVI stream under an .mp4 name."""

    def __init__(self, filename, fourcc, fps, frameSize, isColor=True):
        from yolov3_tensorflow_amd.utils import video_utils
        path = str(filename)
        if int(fourcc) != VideoWriter_fourcc(*'MJPG') or not path.lower().endswith('.avi'):
            path = os.path.splitext(path)[0] + '.avi'
            print('cv2.VideoWriter: no encoder for fourcc %r here; writing Motion-JPEG to %s'
                  % (int(fourcc).to_bytes(4, 'little').decode('latin1'), path), file=sys.stderr)
        self.filename = path
        self._writer = video_utils.MjpegAviWriter(path, fps, frameSize)

    def isOpened(self):
        return self._writer is not None

    def write(self, image):
        self._writer.write(np.asarray(image)[:, :, ::-1])

    def release(self):
        if self._writer is not None:
            self._writer.close()
            self._writer = None
