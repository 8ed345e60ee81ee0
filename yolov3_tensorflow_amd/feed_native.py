# coding: utf-8
"""ctypes binding of liby3feed.so (include/yolo355_feed.h): the per-image CPU work of the feeder in native code.

The numpy / Pillow functions of utils/data_aug.py and utils/data_utils.py DEFINE what this library computes (and pin the
draws and the box arithmetic against the reference); the library produces the same bytes several times faster and in a
single pass (tests/test_feed_native.py holds it to bit equality).  `enabled()` is what the feeder's per-sample code asks:
true unless Y3_FEED_NATIVE=0.  The library is built by yolov3_tensorflow_amd.build (g++, a second and a half); a missing one
is built on first use, and when that is impossible the error is raised, not swallowed.
"""
import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("Y3_FEED_LIB_PATH") or os.path.join(HERE, "csrc", "liby3feed.so")

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = range(5)


class Colour(ctypes.Structure):         # y3f_colour
    _fields_ = [("enabled", c_int32), ("brightness", c_int32), ("hue_on", c_int32), ("hue_delta", c_int32),
                ("sat_gain", c_float), ("val_gain", c_float)]


class Job(ctypes.Structure):            # y3f_job
    _fields_ = [("img1", c_void_p), ("img2", c_void_p), ("h1", c_int32), ("w1", c_int32), ("h2", c_int32), ("w2", c_int32),
                ("lam1", c_float), ("lam2", c_float), ("colour", Colour), ("off_x", c_int32), ("off_y", c_int32),
                ("win_x", c_int32), ("win_y", c_int32), ("win_w", c_int32), ("win_h", c_int32), ("interp", c_int32),
                ("res_w", c_int32), ("res_h", c_int32), ("out_w", c_int32), ("out_h", c_int32), ("pad_x", c_int32),
                ("pad_y", c_int32), ("pad_value", c_int32), ("flip_x", c_int32)]


class DJob(ctypes.Structure):           # y3f_djob: one planned job as the device kernels read it (208 bytes)
    _fields_ = [("img1_off", c_uint64), ("img2_off", c_uint64), ("jitter_off", c_uint64), ("xtab_off", c_uint64),
                ("ytab_off", c_uint64), ("win_off", c_uint64), ("tmp_off", c_uint64),
                ("r1_x0", c_int32), ("r1_y0", c_int32), ("r1_w", c_int32), ("r1_h", c_int32),
                ("r2_x0", c_int32), ("r2_y0", c_int32), ("r2_w", c_int32), ("r2_h", c_int32),
                ("lam1", c_float), ("lam2", c_float), ("has2", c_int32), ("colour_on", c_int32),
                ("img_dx", c_int32), ("img_dy", c_int32),
                ("live_x0", c_int32), ("live_y0", c_int32), ("live_x1", c_int32), ("live_y1", c_int32),
                ("win_w", c_int32), ("win_h", c_int32),
                ("mode", c_int32), ("horizontal", c_int32), ("vertical", c_int32), ("ksize_x", c_int32), ("ksize_y", c_int32),
                ("tmp_y0", c_int32), ("tmp_rows", c_int32),
                ("res_w", c_int32), ("res_h", c_int32), ("out_w", c_int32), ("out_h", c_int32), ("pad_x", c_int32),
                ("pad_y", c_int32), ("pad_value", c_int32), ("flip_x", c_int32), ("reserved", c_int32 * 3)]


DTABLES_BYTES = 3 * 65536 + 65536 + 256 + 3 * 1024       # sizeof(y3f_dtables)

# name -> (restype, argtypes); tests/test_feed_native.py checks this table against the header
PROTOTYPES = {
    "y3f_last_error": (c_char_p, []),
    "y3f_abi_version": (c_int, []),
    "y3f_resize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int]),
    "y3f_rgb_to_hsv": (c_int, [c_void_p, c_void_p, c_size_t]),
    "y3f_hsv_to_rgb": (c_int, [c_void_p, c_void_p, c_size_t]),
    "y3f_colour_distort": (c_int, [c_void_p, c_size_t, POINTER(Colour)]),
    "y3f_sample": (c_int, [POINTER(Job), c_void_p, c_void_p]),
    "y3f_sample_batch": (c_int, [POINTER(Job), c_int, POINTER(c_void_p), POINTER(c_void_p), c_int]),
    "y3f_plan_batch": (c_int, [POINTER(Job), c_int, c_void_p, c_size_t, POINTER(c_size_t), POINTER(c_size_t), c_int]),
    "y3f_device_tables": (c_size_t, [c_void_p, c_size_t]),
    "y3f_crop_candidates": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_double, c_void_p, c_int,
                                    c_int, c_void_p, POINTER(c_int32)]),
}

_lib = None
_lock = threading.Lock()


def enabled():
    return os.environ.get("Y3_FEED_NATIVE", "1") != "0"


def lib():
    """The loaded library.  A missing one is built first (g++, a second and a half; `python -m yolov3_tensorflow_amd.build`
    is what rebuilds a stale one); a failure to build or load is raised."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH) and "Y3_FEED_LIB_PATH" not in os.environ:
                from . import build
                build.build_feed(verbose=False)
            handle = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in PROTOTYPES.items():
                fn = getattr(handle, name)          # AttributeError if the symbol is missing
                fn.restype, fn.argtypes = restype, argtypes
            if handle.y3f_abi_version() != 2:
                raise RuntimeError("liby3feed.so ABI version mismatch: rebuild with `python -m yolov3_tensorflow_amd.build`")
            _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("liby3feed: %s (code %d)" % ((lib().y3f_last_error() or b"").decode(errors="replace"), rc))


def _rgb8(img):
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an HxWx3 uint8 image, got shape %s" % (a.shape,))
    return a


def resize(img, new_width, new_height, interp):
    """cv2.resize(img, (new_width, new_height), interpolation=interp) for an HxWx3 uint8 image (codes 0..4: see the header)."""
    src = _rgb8(img)
    out = np.empty((int(new_height), int(new_width), 3), np.uint8)
    check(lib().y3f_resize(src.ctypes.data, src.shape[0], src.shape[1], out.ctypes.data, out.shape[0], out.shape[1],
                           int(interp)))
    return out


def rgb_to_hsv(img):
    src = _rgb8(img)
    out = np.empty_like(src)
    check(lib().y3f_rgb_to_hsv(src.ctypes.data, out.ctypes.data, src.shape[0] * src.shape[1]))
    return out


def hsv_to_rgb(img):
    src = _rgb8(img)
    out = np.empty_like(src)
    check(lib().y3f_hsv_to_rgb(src.ctypes.data, out.ctypes.data, src.shape[0] * src.shape[1]))
    return out


def make_colour(draws):
    """y3f_colour from data_aug.color_distort_draws' (brightness, hue_delta or None, sat_gain or None, val_gain or None);
    None -> disabled."""
    if draws is None:
        return Colour(0, 0, 0, 0, 1.0, 1.0)
    brightness, hue, sat, val = draws
    return Colour(1, int(brightness), int(hue is not None), int(hue or 0), 1.0 if sat is None else float(sat),
                  1.0 if val is None else float(val))


def colour_distort(img, draws):
    out = _rgb8(img).copy()
    colour = make_colour(draws)
    check(lib().y3f_colour_distort(out.ctypes.data, out.shape[0] * out.shape[1], ctypes.byref(colour)))
    return out


class PixelJob(object):
    """One y3f_job and the arrays its pointers refer to (kept alive with it): what parse_sample hands back instead of pixels
    when the pixel work is left to the device (feeder.Feeder(pixels='gpu'))."""
    __slots__ = ('job', 'img1', 'img2')

    def __init__(self, job, img1, img2):
        self.job, self.img1, self.img2 = job, img1, img2

    @property
    def shape(self):          # (what collate() asks of an image)
        return (self.job.out_h, self.job.out_w, 3)


def make_job(img1, img2=None, lam=1.0, colour=None, offset=(0, 0), window=None, interp=1, resized=None, out_size=None,
             pad=(0, 0), pad_value=128, flip_x=False):
    """The y3f_job of one sample (see the header for the geometry).  img2 / lam: the mix-up partner and img1's weight;
    colour: color_distort_draws' tuple or None; offset = (x, y) of the image on the black canvas; window = (x, y, w, h) on
    the canvas (default: the image); resized = (w, h) the window is resized to (default: out_size); out_size = (w, h)."""
    a = _rgb8(img1)
    b = _rgb8(img2) if img2 is not None else None
    if window is None:
        mh = max(a.shape[0], b.shape[0]) if b is not None else a.shape[0]
        mw = max(a.shape[1], b.shape[1]) if b is not None else a.shape[1]
        window = (0, 0, mw, mh)
    if out_size is None:
        out_size = resized if resized is not None else (window[2], window[3])
    if resized is None:
        resized = out_size
    job = Job()
    job.img1, job.h1, job.w1 = a.ctypes.data, a.shape[0], a.shape[1]
    if b is not None:
        job.img2, job.h2, job.w2 = b.ctypes.data, b.shape[0], b.shape[1]
    job.lam1, job.lam2 = float(lam), 1. - float(lam)
    job.colour = make_colour(colour)
    job.off_x, job.off_y = int(offset[0]), int(offset[1])
    job.win_x, job.win_y, job.win_w, job.win_h = (int(v) for v in window)
    job.interp = int(interp)
    job.res_w, job.res_h = int(resized[0]), int(resized[1])
    job.out_w, job.out_h = int(out_size[0]), int(out_size[1])
    job.pad_x, job.pad_y, job.pad_value, job.flip_x = int(pad[0]), int(pad[1]), int(pad_value), int(bool(flip_x))
    return PixelJob(job, a, b)


def sample(img1, img2=None, lam=1.0, colour=None, offset=(0, 0), window=None, interp=1, resized=None, out_size=None,
           pad=(0, 0), pad_value=128, flip_x=False, out=None, as_float=False):
    """One y3f_sample job (make_job's arguments).  `out`: a preallocated [h, w, 3] array (uint8, or float32 with as_float)
    to write into.  Returns the array written."""
    pj = make_job(img1, img2, lam, colour, offset, window, interp, resized, out_size, pad, pad_value, flip_x)
    job = pj.job
    want = np.float32 if as_float else np.uint8
    if out is None:
        out = np.empty((job.out_h, job.out_w, 3), want)
    elif out.dtype != want or out.shape != (job.out_h, job.out_w, 3) or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous %s array of shape %s" % (want.__name__, (job.out_h, job.out_w, 3)))
    check(lib().y3f_sample(ctypes.byref(job), None if as_float else out.ctypes.data, out.ctypes.data if as_float else None))
    return out


# ---- the device form (include/yolo355_feed.h) ----------------------------------------------------------------------------
def job_array(pixel_jobs):
    arr = (Job * len(pixel_jobs))()
    for i, pj in enumerate(pixel_jobs):
        ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(pj.job), ctypes.sizeof(Job))
    return arr


def plan_sizes(jobs, n):
    """(blob bytes, scratch bytes) y3f_plan_batch wants for these n jobs."""
    blob_bytes, scratch_bytes = c_size_t(0), c_size_t(0)
    check(lib().y3f_plan_batch(jobs, n, None, 0, ctypes.byref(blob_bytes), ctypes.byref(scratch_bytes), 0))
    return blob_bytes.value, scratch_bytes.value


def plan_into(jobs, n, blob_ptr, capacity, threads=0):
    """Writes the batch's blob at blob_ptr if `capacity` suffices; returns (blob bytes, scratch bytes) either way."""
    blob_bytes, scratch_bytes = c_size_t(0), c_size_t(0)
    check(lib().y3f_plan_batch(jobs, n, blob_ptr, capacity, ctypes.byref(blob_bytes), ctypes.byref(scratch_bytes),
                               int(threads)))
    return blob_bytes.value, scratch_bytes.value


def plan_batch(pixel_jobs, threads=0):
    """[PixelJob] -> (blob as a uint8 array, scratch bytes, the DJob records as a ctypes array view of the blob's head)."""
    n = len(pixel_jobs)
    jobs = job_array(pixel_jobs)
    need, _ = plan_sizes(jobs, n)
    blob = np.zeros(max(need, 16), np.uint8)
    _, scratch = plan_into(jobs, n, blob.ctypes.data, blob.size, threads)
    recs = (DJob * n).from_buffer(blob)
    return blob, scratch, recs


def device_tables():
    """y3f_dtables as a uint8 array (uploaded once per device)."""
    out = np.empty(DTABLES_BYTES, np.uint8)
    got = lib().y3f_device_tables(out.ctypes.data, out.size)
    if got != DTABLES_BYTES:
        raise RuntimeError("liby3feed: sizeof(y3f_dtables) is %d, the binding expects %d" % (got, DTABLES_BYTES))
    return out
