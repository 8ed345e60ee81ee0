"""The feeder's pixel work on the device (y3_feed_run, include/yolo355.h; reference
utils/data_utils.py:118-172 after its draws): the three kernels against liby3feed.so's y3f_sample - which is itself
held bit for bit to the numpy / Pillow definition (tests/test_feed_native.py) - on the random jobs of feed_cases.py, on
full-size jobs, and through the Feeder."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import COCO_ANCHORS
from feed_cases import describe, random_case, random_image

pytestmark = pytest.mark.gpu


def _compare(cases, fn, got):
    got = got.cpu().numpy()
    bad = []
    for i, c in enumerate(cases):
        want = fn.sample(as_float=True, **c)
        if not np.array_equal(got[i], want):
            y, x, ch = np.argwhere(got[i] != want)[0]
            bad.append('%s\n   first difference at (y %d, x %d, c %d): %r != %r (%d pixels differ)' %
                       (describe(c), y, x, ch, got[i][y, x, ch] * 255, want[y, x, ch] * 255, int((got[i] != want).any(-1).sum())))
    assert not bad, '%d of %d cases differ:\n%s' % (len(bad), len(cases), '\n'.join(bad[:5]))


@pytest.mark.parametrize('interp', range(5))
def test_kernels_equal_y3f_sample_on_random_jobs(interp):
    from yolov3_tensorflow_amd import feed_native as fn
    from yolov3_tensorflow_amd.feed_device import DevicePixels
    rng = np.random.RandomState(200 + interp)
    dp = DevicePixels()
    for size in ((32, 32), (48, 48), (64, 64)):
        cases = [random_case(rng, out_size=size, interp=interp) for _ in range(150)]
        _compare(cases, fn, dp.run([fn.make_job(**c) for c in cases]))


def test_kernels_equal_y3f_sample_at_training_size():
    """640x480 sources, mix-up partners of another size, 4x expansion, every interpolation, letterbox to 416 and plain
    resize to 608: long filter windows (LANCZOS4 over a 2,500-pixel canvas: 37 taps), large scratch, one batch of 40."""
    from yolov3_tensorflow_amd import feed_native as fn
    from yolov3_tensorflow_amd.feed_device import DevicePixels
    rng = np.random.RandomState(3)
    dp = DevicePixels()
    for out in (416, 608):
        cases = []
        for i in range(40):
            img1 = random_image(rng, 480, 640)
            img2 = random_image(rng, 427, 640) if i % 3 == 0 else None
            mh, mw = 480, 640
            ratio = rng.uniform(1, 4) if i % 2 else 1.0
            cw, ch = int(mw * ratio), int(mh * ratio)
            off = (int(rng.randint(0, cw - mw + 1)), int(rng.randint(0, ch - mh + 1)))
            ww, wh = int(rng.randint(cw // 3, cw + 1)), int(rng.randint(ch // 3, ch + 1))
            window = (int(rng.randint(0, cw - ww + 1)), int(rng.randint(0, ch - wh + 1)), ww, wh)
            if out == 416:
                scale = min(out / ww, out / wh)
                resized = (max(1, int(ww * scale)), max(1, int(wh * scale)))
                pad = ((out - resized[0]) // 2, (out - resized[1]) // 2)
            else:
                resized, pad = (out, out), (0, 0)
            cases.append(dict(img1=img1, img2=img2, lam=float(rng.beta(1.5, 1.5)) if img2 is not None else 1.0,
                              colour=(int(rng.randint(-32, 33)), int(rng.randint(-18, 19)), float(rng.uniform(0.5, 1.5)),
                                      float(rng.uniform(0.5, 1.5))) if i % 4 else None,
                              offset=off, window=window, interp=i % 5, resized=resized, out_size=(out, out), pad=pad,
                              pad_value=128, flip_x=bool(i % 2)))
        _compare(cases, fn, dp.run([fn.make_job(**c) for c in cases], threads=4))


def test_every_colour_goes_through_the_device_jitter_like_the_host():
    """All 2^24 RGB values as one 4096x4096 image through an identity resize, with a jitter and without."""
    from yolov3_tensorflow_amd import feed_native as fn
    from yolov3_tensorflow_amd.feed_device import DevicePixels
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    dp = DevicePixels()
    for colour in ((0, None, None, None), (11, -13, 1.31, 0.77), (-20, 17, 0.6, 1.4)):
        case = dict(img1=img, colour=colour, interp=1, out_size=(4096, 4096))
        got = dp.run([fn.make_job(**case)]).cpu().numpy()[0]
        want = fn.sample(as_float=True, **case)
        assert np.array_equal(got, want), 'jitter %r: %d colours differ' % (colour, int((got != want).any(-1).sum()))


def test_run_reports_bad_arguments():
    from yolov3_tensorflow_amd import _lib, feed_native as fn
    from yolov3_tensorflow_amd import framework as fw
    rng = np.random.RandomState(0)
    pjs = [fn.make_job(**random_case(rng, out_size=(32, 32))) for _ in range(3)]
    blob, scratch, recs = fn.plan_batch(pjs)
    dev_blob = torch.from_numpy(blob).cuda()
    tables = torch.from_numpy(fn.device_tables()).cuda()
    out = torch.empty((3, 32, 32, 3), device='cuda')
    sc = torch.empty(max(scratch, 16), dtype=torch.uint8, device='cuda')
    call = lambda *a: _lib.lib().y3_feed_run(fw.context(), *a)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert call(p(dev_blob), blob.ctypes.data, 3, p(tables), p(sc), sc.numel(), p(out), 32, 32) == 0
    assert call(p(dev_blob), blob.ctypes.data, 3, p(tables), p(sc), sc.numel(), p(out), 48, 32) == _lib.Y3_EINVAL
    if scratch > 16:
        assert call(p(dev_blob), blob.ctypes.data, 3, p(tables), p(sc), scratch - 16, p(out), 32, 32) == _lib.Y3_EINVAL
        assert b'scratch' in _lib.lib().y3_last_error()
    assert call(None, blob.ctypes.data, 3, p(tables), p(sc), sc.numel(), p(out), 32, 32) == _lib.Y3_EINVAL
    torch.cuda.synchronize()


def test_feeder_with_device_pixels_serves_the_batches_of_the_host_path(tmp_path):
    from test_feeder_gpu import _write_set
    from yolov3_tensorflow_amd.feeder import Feeder
    lines = _write_set(tmp_path, 37)
    kw = dict(mode='train', multi_scale=True, use_mix_up=True, num_threads=6, prefetch=3, seed=4, interval=2)
    host = Feeder(lines, 8, 80, [416, 416], COCO_ANCHORS, pixels='host', **kw)
    gpu = Feeder(lines, 8, 80, [416, 416], COCO_ANCHORS, pixels='gpu', **kw)
    served = 0
    for a, b in zip(host.epoch(0), gpu.epoch(0)):
        assert a.image_ids == b.image_ids and a.img_size == b.img_size
        assert b.images.is_cuda and torch.equal(a.images, b.images)
        assert torch.equal(a.boxes, b.boxes) and torch.equal(a.labels, b.labels) and torch.equal(a.counts, b.counts)
        assert all(torch.equal(x, y) for x, y in zip(a.y_true, b.y_true))
        served += len(b.image_ids)
    assert served == 37
    # validation mode: plain letterbox, no draws
    host = Feeder(lines[:9], 4, 80, [416, 416], COCO_ANCHORS, mode='val', num_threads=3, pixels='host')
    gpu = Feeder(lines[:9], 4, 80, [416, 416], COCO_ANCHORS, mode='val', num_threads=3, pixels='gpu')
    for a, b in zip(host.epoch(0), gpu.epoch(0)):
        assert torch.equal(a.images, b.images)
    with pytest.raises(ValueError):
        Feeder(lines, 8, 80, [416, 416], COCO_ANCHORS, backend='process', pixels='gpu')
