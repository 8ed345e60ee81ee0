"""CPU tests of the feeder's device form (include/yolo355_feed.h): y3f_plan_batch's records and blob, and the per-pixel
functions the GPU kernels are made of (csrc/y3_feed_px.h) run on the host (tests/feed_emul.cpp) against y3f_sample -
bit for bit, over every interpolation and the geometry corners.  tests/test_feed_gpu.py repeats the comparison with the
kernels themselves."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from feed_cases import describe, random_case


@pytest.fixture(scope='module')
def fn():
    from yolov3_tensorflow_amd import build, feed_native
    build.build_feed(verbose=False)
    return feed_native


@pytest.fixture(scope='module')
def emul(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('feed_emul') / 'libfeed_emul.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math',
                           os.path.join(ROOT, 'tests', 'feed_emul.cpp'), '-o', out])
    lib = ctypes.CDLL(out)
    lib.y3f_emulate.restype = ctypes.c_int
    lib.y3f_emulate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def run_emulated(fn, emul, cases):
    pjs = [fn.make_job(**c) for c in cases]
    blob, scratch_bytes, recs = fn.plan_batch(pjs)
    tables = fn.device_tables()
    scratch = np.full(max(scratch_bytes, 16), 0xA5, np.uint8)         # poisoned: nothing may be read before it is written
    oh, ow = pjs[0].job.out_h, pjs[0].job.out_w
    out = np.full((len(pjs), oh, ow, 3), np.nan, np.float32)
    assert emul.y3f_emulate(blob.ctypes.data, len(pjs), tables.ctypes.data, scratch.ctypes.data, out.ctypes.data) == 0
    return out, recs


def test_plan_records_are_consistent(fn):
    rng = np.random.RandomState(5)
    cases = [random_case(rng, out_size=(48, 48)) for _ in range(64)]
    pjs = [fn.make_job(**c) for c in cases]
    blob, scratch_bytes, recs = fn.plan_batch(pjs)
    assert ctypes.sizeof(fn.DJob) == 208
    end_blob, end_scratch = 16 * ((len(pjs) * 208 + 15) // 16), 0
    for c, d in zip(cases, recs):
        assert 0 <= d.live_x0 <= d.live_x1 <= d.win_w and 0 <= d.live_y0 <= d.live_y1 <= d.win_h
        lw, lh = d.live_x1 - d.live_x0, d.live_y1 - d.live_y0
        assert (d.r1_w, d.r1_h) == (lw, lh) or d.has2 or lw * lh == 0
        for off, size in ((d.img1_off, d.r1_w * d.r1_h * 3), (d.img2_off, d.r2_w * d.r2_h * 3), (d.jitter_off, 1024 * d.colour_on)):
            assert off % 16 == 0 and off >= end_blob and off + size <= blob.size
        assert d.win_off % 16 == 0 and d.win_off >= end_scratch and d.tmp_off >= d.win_off + lw * lh * 3
        end_scratch = d.tmp_off + d.tmp_rows * d.res_w * 3
        assert end_scratch <= scratch_bytes
        end_blob = d.ytab_off
        assert d.mode in range(5) and (d.mode == 4) == bool(d.horizontal or d.vertical)
        if d.mode == 4 and d.horizontal:
            assert d.live_y0 <= d.tmp_y0 and d.tmp_y0 + d.tmp_rows <= max(d.live_y1, d.tmp_y0)
    # sizing only: no blob written, same numbers
    jobs = fn.job_array(pjs)
    assert fn.plan_sizes(jobs, len(pjs)) == (blob.size, scratch_bytes)
    # too small a buffer is reported, not overrun
    small = np.full(64, 7, np.uint8)
    assert fn.plan_into(jobs, len(pjs), small.ctypes.data, small.size)[0] == blob.size and (small == 7).all()


def test_plan_rejects_what_sample_rejects(fn):
    rng = np.random.RandomState(1)
    a, b = fn.make_job(**random_case(rng, out_size=(32, 32))), fn.make_job(**random_case(rng, out_size=(48, 48)))
    with pytest.raises(RuntimeError, match='job 1 writes'):
        fn.plan_batch([a, b])
    a.job.interp = 9
    with pytest.raises(RuntimeError, match='interpolation code 9'):
        fn.plan_batch([a])
    a.job.interp, a.job.win_w = 1, 0
    with pytest.raises(RuntimeError, match='empty window'):
        fn.plan_batch([a])


@pytest.mark.parametrize('interp', range(5))
def test_device_functions_equal_y3f_sample(fn, emul, interp):
    rng = np.random.RandomState(100 + interp)
    bad = []
    for size in ((32, 32), (48, 48), (64, 64)):
        cases = [random_case(rng, out_size=size, interp=interp) for _ in range(120)]
        got, recs = run_emulated(fn, emul, cases)
        for i, c in enumerate(cases):
            want = fn.sample(as_float=True, **c)
            if not np.array_equal(got[i], want):
                y, x, ch = np.argwhere(got[i] != want)[0]
                bad.append('%s\n   mode %d first difference at (y %d, x %d, c %d): %r != %r' %
                           (describe(c), recs[i].mode, y, x, ch, got[i][y, x, ch] * 255, want[y, x, ch] * 255))
    assert not bad, '%d of 360 cases differ:\n%s' % (len(bad), '\n'.join(bad[:5]))


def test_modes_are_all_exercised(fn):
    rng = np.random.RandomState(7)
    seen = set()
    for _ in range(600):
        pj = fn.make_job(**random_case(rng, out_size=(48, 48)))
        _, _, recs = fn.plan_batch([pj])
        d = recs[0]
        seen.add((d.mode, d.horizontal, d.vertical, d.has2, d.colour_on, (d.live_x1 - d.live_x0) * (d.live_y1 - d.live_y0) == 0))
    modes = {m for m, *_ in seen}
    assert modes == {0, 1, 2, 3, 4}
    assert {(h, v) for m, h, v, *_ in seen if m == 4} == {(1, 1), (1, 0), (0, 1)}
    assert any(empty for *_, empty in seen) and any(h2 for _, _, _, h2, _, _ in seen)
