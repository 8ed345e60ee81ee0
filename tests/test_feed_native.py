"""liby3feed.so (include/yolo355_feed.h): the feeder's native pixel path against what DEFINES it - the numpy / Pillow
functions of utils/data_aug.py and utils/data_utils.py (themselves pinned draw for draw against the reference module:
tests/test_feeder_cpu.py).  Everything here is byte / integer work, so the bar is bit equality.  No GPU."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'yolo355_feed.h')


@pytest.fixture(scope='module')
def fn():
    from yolov3_tensorflow_amd import build, feed_native
    build.build_feed(verbose=False)
    feed_native.lib()
    return feed_native


def _smooth(rng, h, w):
    from PIL import Image
    small = rng.randint(0, 256, (max(2, h // 9), max(2, w // 9), 3)).astype(np.uint8)
    return np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC))


def test_library_exports_what_the_header_declares(fn):
    text = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    declared = sorted(set(re.findall(r'\b(y3f_[a-z0-9_]+)\s*\(', text)))
    assert declared == sorted(fn.PROTOTYPES)
    for name in declared:
        assert hasattr(fn.lib(), name), 'liby3feed.so does not export %s' % name
    assert fn.lib().y3f_abi_version() == 2
    # the ctypes mirrors have the header's field order and sizes (all 4-byte fields after the two pointers)
    assert ctypes.sizeof(fn.Colour) == 24 and ctypes.sizeof(fn.Job) == 128 and ctypes.sizeof(fn.DJob) == 208      # (static_assert'ed in y3_feed.cpp)
    # errors come back as codes with a message, not as crashes
    out = np.empty((4, 4, 3), np.uint8)
    assert fn.lib().y3f_resize(out.ctypes.data, 4, 4, out.ctypes.data, 4, 4, 9) == -1
    assert b'interpolation' in fn.lib().y3f_last_error()
    with pytest.raises(RuntimeError, match='does not fit'):
        fn.sample(out, resized=(8, 8), out_size=(4, 4))
    with pytest.raises(RuntimeError, match='out of range'):
        fn.sample(out, window=(2 ** 30, 0, 4, 4), out_size=(4, 4))


def test_hsv_conversions_equal_pillows_on_every_colour(fn):
    from PIL import Image
    axis = np.arange(256, dtype=np.uint8)
    for first in range(256):
        cube = np.stack(np.meshgrid(np.full(1, first, np.uint8), axis, axis, indexing='ij'), -1).reshape(256, 256, 3).copy()
        np.testing.assert_array_equal(fn.rgb_to_hsv(cube), np.asarray(Image.fromarray(cube).convert('HSV')))
        np.testing.assert_array_equal(fn.hsv_to_rgb(cube), np.asarray(Image.fromarray(cube, 'HSV').convert('RGB')))


def test_colour_jitter_equals_the_numpy_definition_on_every_colour(fn):
    """The per-pixel loop runs on tables (hue by channel differences, saturation by (max, min)) and a rounding shortcut;
    this is the all-inputs comparison its comments refer to: 2^24 colours x jitters that exercise every branch."""
    from yolov3_tensorflow_amd.utils import data_aug
    axis = np.arange(256, dtype=np.uint8)
    jitters = [(0, None, None, None), (0, 7, None, None), (0, -18, 1.37, 0.61), (25, None, 0.5, 1.5),
               (-31, 17, 1.4999, 0.5001)]
    for first in range(256):
        cube = np.stack(np.meshgrid(np.full(1, first, np.uint8), axis, axis, indexing='ij'), -1).reshape(256, 256, 3).copy()
        for draws in (jitters if first % 32 == 0 else jitters[:2]):
            np.testing.assert_array_equal(fn.colour_distort(cube, draws), data_aug.apply_color_distort(cube, draws))
    # and with drawn amounts on an image
    rng = np.random.RandomState(4)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    for seed in range(40):
        draws = data_aug.color_distort_draws(rng=np.random.RandomState(seed))
        np.testing.assert_array_equal(fn.colour_distort(img, draws), data_aug.apply_color_distort(img, draws))


SIZES = [(480, 640, 416, 416), (375, 500, 320, 320), (200, 300, 608, 608), (50, 37, 416, 416), (832, 832, 416, 416),
         (416, 416, 416, 416), (100, 416, 416, 33), (1, 1, 5, 7), (3, 1000, 416, 416), (333, 416, 200, 416),
         (1080, 1920, 416, 234), (7, 5, 3, 2), (64, 96, 64, 48), (96, 64, 48, 64)]


@pytest.mark.parametrize('interp', range(5))
def test_resize_equals_the_numpy_and_pillow_definitions(fn, interp, monkeypatch):
    """Codes 0 / 1: OpenCV's INTER_NEAREST / INTER_LINEAR as utils.data_utils restates them; 2 / 3 / 4: Pillow's BICUBIC /
    BOX / LANCZOS (what data_aug._resize_any calls).  Up- and down-scaling, one-axis-only, degenerate and 2x sizes."""
    from yolov3_tensorflow_amd.utils import data_aug
    rng = np.random.RandomState(interp)
    for h, w, nh, nw in SIZES:
        for img in (rng.randint(0, 256, (h, w, 3)).astype(np.uint8), _smooth(rng, h, w)):
            monkeypatch.setenv('Y3_FEED_NATIVE', '0')              # the definition: numpy restatements / Pillow
            want = data_aug._resize_any(img, nw, nh, interp)
            monkeypatch.delenv('Y3_FEED_NATIVE')
            np.testing.assert_array_equal(fn.resize(img, nw, nh, interp), want, err_msg=str((h, w, nh, nw)))


def test_crop_search_draws_from_pythons_generator_exactly(fn):
    """y3f_crop_candidates runs the trial loop on random.Random's own Mersenne Twister state: the same windows as the
    Python loop and the generator left at the same position - checked by what it draws NEXT."""
    from yolov3_tensorflow_amd.utils import data_aug
    rng = np.random.RandomState(0)
    bands = [(-np.inf if lo is None else lo, np.inf if hi is None else hi) for lo, hi in data_aug._DEFAULT_IOU_BANDS]
    for seed in range(120):
        w, h = int(rng.randint(40, 900)), int(rng.randint(40, 700))
        n = int(rng.randint(0, 6)) if seed % 7 else 0
        xy = rng.uniform(0, [w * 0.7, h * 0.7], (n, 2))
        boxes = np.concatenate([xy, xy + rng.uniform(4, [w * 0.3, h * 0.3], (n, 2))], 1).astype(np.float32)
        rows = [tuple(float(v) for v in b) for b in boxes]
        args = (rows, w, h, 0.3, 1, 2, bands if seed % 5 else bands[:2] + [(0.2, 0.6)], 50 if seed % 3 else 7)
        a, b = random.Random(seed), random.Random(seed)
        if seed % 11 == 0:          # a generator in the middle of its 624-word block, and one about to refill it
            for g in (a, b):
                [g.random() for _ in range(300 + seed)]
        assert data_aug._candidate_windows_native(a, *args) == data_aug._candidate_windows(b, *args)
        assert a.getstate() == b.getstate() and a.random() == b.random()
    # the module-level generator (what the reference's functions use) works the same way
    random.seed(5)
    first = data_aug._candidate_windows_native(random, rows, w, h, 0.3, 1, 2, bands, 50)
    random.seed(5)
    assert data_aug._candidate_windows(random, rows, w, h, 0.3, 1, 2, bands, 50) == first


def _write_images(tmp_path, n=10):
    from PIL import Image
    rng = np.random.RandomState(11)
    lines = []
    for i in range(n):
        w, h = int(rng.randint(120, 420)), int(rng.randint(90, 330))
        path = str(tmp_path / ('n%d.jpg' % i))
        Image.fromarray(_smooth(rng, h, w)).save(path, quality=88)
        parts = ['%d' % i, path, '%d' % w, '%d' % h]
        for _ in range(int(rng.randint(1, 5))):
            x0, y0 = rng.uniform(0, w * 0.5), rng.uniform(0, h * 0.5)
            parts += ['%d' % rng.randint(0, 80), '%.1f' % x0, '%.1f' % y0, '%.1f' % (x0 + rng.uniform(10, w * 0.45)),
                      '%.1f' % (y0 + rng.uniform(10, h * 0.45))]
        lines.append(' '.join(parts))
    return lines


def test_whole_sample_equals_the_numpy_chain(fn, tmp_path, monkeypatch):
    """parse_sample with the native pixel path against the same call on the numpy / Pillow chain (Y3_FEED_NATIVE=0): same
    image bytes (uint8 and float32 forms), same boxes, same labels, and both generators left at the same position - train
    and val, plain and letterbox, single images and mix-up pairs of different sizes, all five interpolations."""
    from yolov3_tensorflow_amd.utils import data_utils
    lines = _write_images(tmp_path)

    def run(native, i, pair, mode, letterbox, as_u8):
        monkeypatch.setenv('Y3_FEED_NATIVE', '1' if native else '0')
        rng, prng = np.random.RandomState(i), random.Random(i)
        line = [lines[i % len(lines)], lines[(i * 7 + 3) % len(lines)]] if pair else lines[i % len(lines)]
        out = data_utils.parse_sample(line, [160, 128] if i % 2 else [128, 160], mode, letterbox, rng=rng, prng=prng,
                                      as_uint8=as_u8)
        return out, rng.uniform(), prng.random()

    for i in range(240):
        for pair in (False, True):
            mode, letterbox, as_u8 = ('train' if i % 5 else 'val'), bool((i // 2) % 2), bool(i % 3)
            (a, ra, pa), (b, rb, pb) = run(True, i, pair, mode, letterbox, as_u8), run(False, i, pair, mode, letterbox, as_u8)
            assert (ra, pa) == (rb, pb) and a[0] == b[0]
            for x, y in zip(a[1:], b[1:]):
                assert x.dtype == y.dtype and x.shape == y.shape
                np.testing.assert_array_equal(x, y, err_msg=str((i, pair, mode, letterbox)))


def test_sample_batch_runs_jobs_on_its_own_threads(fn):
    rng = np.random.RandomState(2)
    imgs = [_smooth(rng, int(rng.randint(60, 200)), int(rng.randint(60, 200))) for _ in range(12)]
    jobs = (fn.Job * len(imgs))()
    outs = [np.empty((96, 128, 3), np.float32) for _ in imgs]
    for i, (img, job) in enumerate(zip(imgs, jobs)):
        job.img1, job.h1, job.w1 = img.ctypes.data, img.shape[0], img.shape[1]
        job.colour = fn.make_colour((i - 6, i - 5, 1.1, 0.9))
        job.win_w, job.win_h, job.interp = img.shape[1], img.shape[0], i % 5
        job.res_w = job.out_w = 128
        job.res_h = job.out_h = 96
        job.flip_x = i % 2
    ptrs = (ctypes.c_void_p * len(imgs))(*[o.ctypes.data for o in outs])
    fn.check(fn.lib().y3f_sample_batch(jobs, len(imgs), None, ptrs, 4))
    for i, img in enumerate(imgs):
        want = fn.sample(img, colour=(i - 6, i - 5, 1.1, 0.9), interp=i % 5, out_size=(128, 96), flip_x=i % 2, as_float=True)
        np.testing.assert_array_equal(outs[i], want)
    jobs[7].win_w = 0                    # one bad job: its error comes back, the others still ran
    assert fn.lib().y3f_sample_batch(jobs, len(imgs), None, ptrs, 3) == -1
    assert b'job 7' in fn.lib().y3f_last_error()


def test_sample_geometry_against_a_numpy_construction(fn):
    """y3f_sample's contract on its own (include/yolo355_feed.h): image (or mix-up blend) at an offset on an unbounded
    black canvas, any window of that canvas - inside, overlapping or entirely off the image -, one resize, placement on a
    padded field, mirror, /255."""
    from yolov3_tensorflow_amd.utils import data_aug
    rng = np.random.RandomState(8)
    for case in range(60):
        h1, w1 = int(rng.randint(5, 60)), int(rng.randint(5, 60))
        img1 = rng.randint(0, 256, (h1, w1, 3)).astype(np.uint8)
        img2, lam = None, 1.0
        if case % 3 == 0:
            img2 = rng.randint(0, 256, (int(rng.randint(5, 60)), int(rng.randint(5, 60)), 3)).astype(np.uint8)
            lam = float(rng.beta(1.5, 1.5))
        src = img1 if img2 is None else data_aug.blend(img1, img2, lam)
        colour = None if case % 4 else data_aug.color_distort_draws(rng=np.random.RandomState(case))
        if colour is not None:
            src = data_aug.apply_color_distort(src, colour)
        off = (int(rng.randint(-20, 40)), int(rng.randint(-20, 40)))
        win = (int(rng.randint(-30, 60)), int(rng.randint(-30, 60)), int(rng.randint(1, 90)), int(rng.randint(1, 90)))
        # the numpy construction: a canvas large enough for everything, shifted so that no index is negative
        shift = 64
        canvas = np.zeros((256, 256, 3), np.uint8)
        canvas[off[1] + shift:off[1] + shift + src.shape[0], off[0] + shift:off[0] + shift + src.shape[1]] = src
        window = canvas[win[1] + shift:win[1] + shift + win[3], win[0] + shift:win[0] + shift + win[2]]
        interp = case % 5
        res = (int(rng.randint(1, 50)), int(rng.randint(1, 50)))
        out = (res[0] + int(rng.randint(0, 9)), res[1] + int(rng.randint(0, 9)))
        pad = (int(rng.randint(0, out[0] - res[0] + 1)), int(rng.randint(0, out[1] - res[1] + 1)))
        flip = bool(case % 2)
        want = np.full((out[1], out[0], 3), 77, np.uint8)
        want[pad[1]:pad[1] + res[1], pad[0]:pad[0] + res[0]] = fn.resize(window, res[0], res[1], interp)
        if flip:
            want = want[:, ::-1]
        got = fn.sample(img1, img2, lam, colour, off, win, interp, res, out, pad, 77, flip)
        np.testing.assert_array_equal(got, want, err_msg='case %d' % case)
        as_float = fn.sample(img1, img2, lam, colour, off, win, interp, res, out, pad, 77, flip, as_float=True)
        np.testing.assert_array_equal(as_float, want.astype(np.float32) / np.float32(255.))


def test_a_c_host_links_the_library_and_gets_the_same_bytes(fn, tmp_path):
    """examples/feed_sample.c: a plain C program against include/yolo355_feed.h (no Python in the process) runs one job;
    the Python binding, given the same job, produces the same image."""
    import shutil
    import subprocess
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        pytest.skip('no C compiler on this machine')
    csrc = os.path.join(ROOT, 'yolov3_tensorflow_amd', 'csrc')
    exe = str(tmp_path / 'feed_sample')
    subprocess.check_call([cc, '-O2', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'feed_sample.c'),
                           '-o', exe, '-L', csrc, '-ly3feed', '-Wl,-rpath,' + csrc])
    rng = np.random.RandomState(5)
    img = _smooth(rng, 150, 210)
    src, dst = str(tmp_path / 'in.ppm'), str(tmp_path / 'out.ppm')
    with open(src, 'wb') as f:
        f.write(b'P6\n210 150\n255\n' + img.tobytes())
    text = subprocess.check_output([exe, src, dst]).decode()
    raw = open(dst, 'rb').read()
    assert raw.startswith(b'P6\n416 416\n255\n')
    got = np.frombuffer(raw[len(b'P6\n416 416\n255\n'):], np.uint8).reshape(416, 416, 3)
    want = fn.sample(img, colour=(9, -7, 1.25, 0.9), offset=(105, 75), window=(52, 37, 210, 150), interp=4,
                     out_size=(416, 416), flip_x=True)
    np.testing.assert_array_equal(got, want)
    assert text.startswith('abi 2;') and ('%.6f' % (want[415, 0, 0] / np.float32(255.))) in text
