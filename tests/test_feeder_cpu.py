"""Host logic of the feeder: the augmentation box arithmetic against goldens produced by the
REFERENCE's own utils/data_aug.py (tests/golden/make_aug_golden.py: same process-global generators, same seeds -> the same
draws), the multi-scale size sequence of get_batch_data, the deterministic batch plan (mix-up pairing, rank sharding)."""
import os
import random

import numpy as np
import pytest

from yolov3_tensorflow_amd.utils import data_aug

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_aug_goldens.npz')


@pytest.fixture(scope='module')
def g():
    return np.load(GOLD)


def test_bbox_iou_and_crop_equal_the_reference(g):
    np.testing.assert_array_equal(data_aug.bbox_iou(g['iou_a'], g['iou_b']), g['iou_out'])
    for i in range(3):
        got = data_aug.bbox_crop(g['crop_in'], tuple(int(v) for v in g['crop_box_%d' % i]),
                                 allow_outside_center=bool(g['crop_aoc_%d' % i]))
        np.testing.assert_array_equal(got, g['crop_out_%d' % i])


def test_random_crop_expand_flip_mixup_reproduce_the_reference_draw_for_draw(g):
    """Handed the process-global generators under the reference's seeds, the restated functions consume the same random
    numbers in the same order and return the same boxes / crops / canvases."""
    for seed in g['rc_seeds']:
        seed = int(seed)
        random.seed(seed)
        np.random.seed(seed)
        nb, crop = data_aug.random_crop_with_constraints(g['rc_in_%d' % seed].copy(), (500, 375))
        np.testing.assert_array_equal(np.array(crop, np.int64), g['rc_crop_%d' % seed])
        np.testing.assert_array_equal(nb, g['rc_out_%d' % seed])
    for seed in range(6):
        img, box = g['ex_img_%d' % seed], g['ex_in_%d' % seed]
        random.seed(seed)
        np.random.seed(seed)
        e_img, e_box = data_aug.random_expand(img.copy(), box.copy(), 4)
        assert tuple(e_img.shape) == tuple(g['ex_shape_%d' % seed]) and int(e_img.astype(np.int64).sum()) == int(g['ex_sum_%d' % seed])
        np.testing.assert_array_equal(e_box, g['ex_out_%d' % seed])
        random.seed(seed)
        np.random.seed(seed)
        f_img, f_box = data_aug.random_flip(img.copy(), box.copy(), px=0.5, py=0.3)
        np.testing.assert_array_equal(f_box, g['fl_out_%d' % seed])
        np.testing.assert_array_equal(f_img, g['fl_img_%d' % seed])
        np.random.seed(seed)
        m_img, m_box = data_aug.mix_up(img, g['mx_img2_%d' % seed], box[:, :4], g['mx_in2_%d' % seed])
        np.testing.assert_array_equal(m_img, g['mx_img_%d' % seed])
        np.testing.assert_array_equal(m_box, g['mx_out_%d' % seed])


def test_multi_scale_sizes_follow_get_batch_data(g):
    from yolov3_tensorflow_amd.utils.data_utils import multi_scale_size
    got = np.array([multi_scale_size(c, 10) for c in range(120)], np.int64)
    np.testing.assert_array_equal(got, g['ms_sizes'])
    assert got.min() == 320 and got.max() <= 608                 # range(10, 20) * 32: never 640


def test_hsv_round_trip_and_colour_jitter_range():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (24, 31, 3)).astype(np.uint8)
    back = data_aug.hsv_to_rgb_u8(data_aug.rgb_to_hsv_u8(img))
    assert np.abs(back.astype(np.int32) - img.astype(np.int32)).max() <= 4      # 8-bit HSV quantisation (H in 2-degree steps)
    hsv = data_aug.rgb_to_hsv_u8(np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]], np.uint8))
    np.testing.assert_array_equal(hsv[0, :, 0], [0, 60, 120, 0])               # OpenCV's H / 2 convention
    out = data_aug.random_color_distort(img, rng=np.random.RandomState(3))
    assert out.dtype == np.uint8 and out.shape == img.shape


def _write_set(tmp_path, n=7, classes=3):
    from PIL import Image
    rng = np.random.RandomState(5)
    lines = []
    for i in range(n):
        w, h = int(rng.randint(80, 160)), int(rng.randint(80, 160))
        path = str(tmp_path / ('img_%d.jpg' % i))
        Image.fromarray(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(path, quality=90)
        k = int(rng.randint(1, 4))
        parts = ['%d' % i, path, '%d' % w, '%d' % h]
        for _ in range(k):
            x0, y0 = rng.uniform(0, w * 0.5), rng.uniform(0, h * 0.5)
            parts += ['%d' % rng.randint(0, classes), '%.1f' % x0, '%.1f' % y0, '%.1f' % (x0 + rng.uniform(10, w * 0.4)),
                      '%.1f' % (y0 + rng.uniform(10, h * 0.4))]
        lines.append(' '.join(parts))
    return lines


def test_train_samples_keep_valid_boxes_and_are_reproducible(tmp_path):
    from yolov3_tensorflow_amd.utils.data_utils import parse_sample
    lines = _write_set(tmp_path)
    for j, line in enumerate(lines):
        a = parse_sample(line, [96, 64], 'train', True, rng=np.random.RandomState(j), prng=random.Random(j))
        b = parse_sample(line, [96, 64], 'train', True, rng=np.random.RandomState(j), prng=random.Random(j))
        idx, img, boxes, labels = a
        assert idx == j and img.shape == (64, 96, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0
        assert boxes.shape[1] == 5 and 1 <= len(boxes) <= len(labels)      # the crop drops boxes, never labels (quirk)
        assert (boxes[:, 0] <= boxes[:, 2]).all() and (boxes[:, 1] <= boxes[:, 3]).all()
        assert boxes[:, :4].min() >= -1e-3 and boxes[:, 2].max() <= 96 + 1e-3 and boxes[:, 3].max() <= 64 + 1e-3
        np.testing.assert_array_equal(img, b[1])
        np.testing.assert_array_equal(boxes, b[2])
    # a mix-up pair: both images' boxes, weights summing to one per pair
    idx, img, boxes, labels = parse_sample([lines[0], lines[1]], [96, 64], 'val', False, rng=np.random.RandomState(9))
    w = np.unique(np.round(boxes[:, 4], 6))
    assert len(w) <= 2 and abs(w.sum() - 1.0) < 1e-5 if len(w) == 2 else True


def test_feeder_plan_is_deterministic_sharded_and_paired(tmp_path):
    from yolov3_tensorflow_amd.feeder import Feeder
    lines = _write_set(tmp_path, n=10)
    anchors = np.arange(18, dtype=np.float32).reshape(9, 2) + 5
    mk = lambda **kw: Feeder(lines, 4, 3, [64, 64], anchors, mode='train', multi_scale=True, use_mix_up=True, seed=3, **kw)
    a, b = mk()._plan(2), mk()._plan(2)
    assert [(x[0], x[1]) for x in a] == [(x[0], x[1]) for x in b] and len(a) == 3
    assert all(str(x[2]) == str(y[2]) for x, y in zip(a, b))
    assert mk()._plan(1)[0][2] != a[0][2] or len(lines) < 3       # another epoch, another order
    r0, r1 = mk(rank=0, world=2)._plan(2), mk(rank=1, world=2)._plan(2)
    for whole, p0, p1 in zip(a, r0, r1):
        assert whole[1] == p0[1] == p1[1]                         # same image size on every rank
        assert whole[2][0::2] == p0[2] and (whole[2][1::2] or whole[2][:1]) == p1[2]
    assert any(isinstance(l, list) for _, _, ls in a for l in ls)  # some lines were paired for mix-up
    sizes = {tuple(x[1]) for x in a}
    assert all(s[0] % 32 == 0 and 320 <= s[0] <= 608 for s in sizes)
    v = Feeder(lines, 4, 3, [64, 64], anchors, mode='val')._plan(0)
    assert [l for _, _, ls in v for l in ls] == lines             # validation: file order, no pairing, fixed size


def test_worker_pool_is_shared_started_from_a_forkserver_and_reproducible(tmp_path):
    """The decode / augment workers come from ONE forkserver pool per worker count (a fork of the training process copies
    its pinned pages: 93 s for four workers at 8 GB pinned, tools/feeder_diag.py); what they return equals the in-process
    call with the same random key."""
    from yolov3_tensorflow_amd import feeder
    lines = _write_set(tmp_path)
    pool = feeder._shared_process_pool(3)
    assert feeder._shared_process_pool(3) is pool
    assert pool._mp_context.get_start_method() == 'forkserver'
    jobs = [(line, [96, 64], 'train', True, 1000 + j) for j, line in enumerate(lines)] + \
           [([lines[0], lines[1]], [96, 64], 'train', False, 77)]
    got = [f.result(timeout=120) for f in [pool.submit(feeder._worker_sample, job) for job in jobs]]
    for job, g in zip(jobs, got):
        want = feeder._worker_sample(job)
        assert g[0] == want[0] and g[1].dtype == np.uint8
        for a, b in zip(g[1:], want[1:]):
            np.testing.assert_array_equal(a, b)
    f1 = feeder.Feeder(lines, 2, 80, [96, 64], np.arange(18, dtype=np.float32), num_threads=3, backend='process')
    f2 = feeder.Feeder(lines, 2, 80, [96, 64], np.arange(18, dtype=np.float32), mode='val', num_threads=3,
                       backend='process')
    assert f1._executor() is pool and f2._executor() is pool
    f1.close()
    assert feeder._shared_process_pool(3) is pool          # closing a feeder leaves the shared workers alone
    # the default: threads (the pixel work is native code that releases the GIL), each writing its float32 slot in place
    f3 = feeder.Feeder(lines, 2, 80, [96, 64], np.arange(18, dtype=np.float32), num_threads=3)
    assert f3.backend == 'thread' and f3._executor() is not pool
    slot = np.empty((64, 96, 3), np.float32)
    got = f3._executor().submit(feeder._worker_sample, jobs[0], slot).result(timeout=120)
    want = feeder._worker_sample(jobs[0])
    assert got[1] is slot
    np.testing.assert_array_equal(slot, want[1].astype(np.float32) / np.float32(255.))
    f3.close()


def test_batch_workers_of_get_batch_data(tmp_path, monkeypatch):
    """utils.data_utils.BATCH_WORKERS (the reference's num_parallel_calls, handed over by the compat tf.data shim): 'val'
    batches do not depend on it; 'train' batches on worker threads are reproducible from the global numpy seed."""
    import threading
    from yolov3_tensorflow_amd.utils import data_utils
    lines = _write_set(tmp_path)
    one = data_utils._map_samples(lines, [96, 64], 'val', True)
    monkeypatch.setattr(data_utils, 'BATCH_WORKERS', 4)
    seen = set()
    real = data_utils.parse_sample

    def spy(*a, **k):
        seen.add(threading.current_thread().name)
        return real(*a, **k)
    monkeypatch.setattr(data_utils, 'parse_sample', spy)
    four = data_utils._map_samples(lines, [96, 64], 'val', True)
    assert all(n.startswith('y3-batch') for n in seen) and len(one) == len(four) == len(lines)
    for a, b in zip(one, four):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
    np.random.seed(11)
    t1 = data_utils._map_samples(lines, [96, 64], 'train', False)
    np.random.seed(11)
    t2 = data_utils._map_samples(lines, [96, 64], 'train', False)
    for a, b in zip(t1, t2):
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
    assert any(not np.array_equal(a[1], b[1]) for a, b in zip(t1, one))         # augmented: not the plain resize
    # the shim's map() is what raises the setting
    from yolov3_tensorflow_amd import compat
    compat.install()
    import tensorflow as tf
    monkeypatch.setattr(data_utils, 'BATCH_WORKERS', 1)
    tf.data.TextLineDataset(str(tmp_path / 'none.txt')).batch(2).map(lambda x: x, num_parallel_calls=6).prefetch(3)
    assert data_utils.BATCH_WORKERS == 6


def test_worker_processes_fill_shared_batch_buffers(tmp_path):
    """backend='process' with the native pixel path: a worker process writes its float32 image straight into its slot of a
    batch buffer the parent keeps in /dev/shm (page-locked for the device when there is one); only boxes and labels travel
    through the pipe.  The slot then holds exactly what the in-process call produces."""
    import os
    from yolov3_tensorflow_amd import feeder
    if not os.path.isdir('/dev/shm'):
        import pytest
        pytest.skip('no /dev/shm on this machine')
    lines = _write_set(tmp_path)
    shared = feeder._SharedBuffers()
    entry = shared.take((3, 64, 96, 3))
    assert entry is not None and os.path.exists(entry['path']) and entry['array'].shape == (3, 64, 96, 3)
    pool = feeder._shared_process_pool(3)
    jobs = [(lines[0], [96, 64], 'train', True, 501), ([lines[1], lines[2]], [96, 64], 'train', False, 502),
            (lines[3], [96, 64], 'val', True, 503)]
    got = [f.result(timeout=120) for f in [pool.submit(feeder._worker_sample_shared, job, entry['path'], (3, 64, 96, 3), j)
                                           for j, job in enumerate(jobs)]]
    for j, (job, g) in enumerate(zip(jobs, got)):
        want = feeder._worker_sample(job, out=np.empty((64, 96, 3), np.float32))
        assert g[0] == want[0] and g[1] is None
        np.testing.assert_array_equal(entry['array'][j], want[1])
        np.testing.assert_array_equal(g[2], want[2])
        np.testing.assert_array_equal(g[3], want[3])
    path = entry['path']
    shared.close()
    assert not os.path.exists(path) and shared.take((1, 2, 2, 3)) is not None       # (usable again after close)
    shared.close()


@pytest.mark.parametrize('native', ['1', '0'])
def test_fix_crop_labels_keeps_each_box_with_its_class(tmp_path, monkeypatch, native):
    """Default: the reference's pairing (the crop filters boxes, not labels - utils/data_utils.py:150-153): after a crop that
    dropped an earlier box the survivors carry their predecessors' classes.  Y3_FIX_CROP_LABELS=1 (train.py
    --fix_crop_labels): the class rides through the chain with its box.  Same draws, same pixels, same boxes either way."""
    from PIL import Image
    from yolov3_tensorflow_amd.utils.data_utils import parse_sample, collate
    monkeypatch.setenv('Y3_FEED_NATIVE', native)
    path = str(tmp_path / 'two.jpg')
    Image.fromarray(np.random.RandomState(1).randint(0, 256, (120, 160, 3)).astype(np.uint8)).save(path, quality=90)
    line = '0 %s 160 120 0 2 2 30 30 2 100 70 150 110' % path        # class 0 top-left, class 2 bottom-right

    def crop_around_second_box(bbox, size, **kw):                    # (stands in for the random window search)
        x0, y0, x1, y1 = [int(v) for v in bbox[1, :4]]
        win = (x0 - 4, y0 - 4, x1 - x0 + 8, y1 - y0 + 8)
        assert win[0] + win[2] <= size[0] and win[1] + win[3] <= size[1]
        return data_aug.bbox_crop(bbox, win, allow_outside_center=False), win

    monkeypatch.setattr(data_aug, 'random_crop_with_constraints', crop_around_second_box)
    for seed in range(6):                                            # with and without expansion / flip
        got = {}
        for flag in ('0', '1'):
            monkeypatch.setenv('Y3_FIX_CROP_LABELS', flag)
            got[flag] = parse_sample(line, [96, 64], 'train', True, rng=np.random.RandomState(seed), prng=random.Random(seed))
        (_, img0, b0, l0), (_, img1, b1, l1) = got['0'], got['1']
        np.testing.assert_array_equal(img0, img1)
        np.testing.assert_array_equal(b0, b1)
        assert b1.shape == (1, 5) and l1.dtype == np.int64
        assert list(l0) == [0, 2]                                    # never filtered: collate pairs the survivor with class 0
        assert list(l1) == [2]
        assert collate([got['0']])[3][0, 0] == 0 and collate([got['1']])[3][0, 0] == 2
    monkeypatch.setenv('Y3_FIX_CROP_LABELS', '1')                    # 'val' mode has no crop: nothing to carry
    _, _, bv, lv = parse_sample(line, [96, 64], 'val', True)
    assert bv.shape == (2, 5) and list(lv) == [0, 2]
