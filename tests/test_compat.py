"""The script-compatibility layer (yolov3_tensorflow_amd/compat: `tensorflow`, `cv2`, `model`, `utils` shims + runner).

  * no GPU, needs /root/reference (skipped where it does not exist, e.g. on the GPU box): the reference's
    test_single_image.py and convert_weight.py run BYTE-UNCHANGED through the runner in dry-run mode - every symbol
    they touch exists and behaves (graph building, Session.run plumbing, image I/O and drawing through the cv2 shim);
  * no GPU: the cv2 shim against known values; the deferred graph's feed / fetch / memoisation rules;
  * GPU: TF-1 style twins of the two scripts (tests/compat_scripts, same symbols) run for real - darknet file ->
    checkpoint -> detections on the demo image - and reproduce the golden detections of tests/golden/messi_config1.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
ANCHORS_TXT = os.path.join(ROOT, 'data', 'yolo_anchors.txt')


def _run(script, args, cwd, extra_env=None, timeout=600):
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, '-m', 'yolov3_tensorflow_amd.compat.run', script] + list(args), cwd=cwd,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return r.returncode, r.stdout.decode(errors='replace')


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is not on this machine')
def test_reference_demo_script_runs_unchanged_in_dry_run(tmp_path):
    os.makedirs(tmp_path / 'data')
    for f in ('yolo_anchors.txt', 'coco.names'):
        with open(os.path.join(ROOT, 'data', f)) as src, open(tmp_path / 'data' / f, 'w') as dst:
            dst.write(src.read())
    rc, out = _run(os.path.join(REF, 'test_single_image.py'), [os.path.join(HERE, 'golden', 'messi.jpg')], str(tmp_path),
                   {'Y3_COMPAT_DRY_RUN': '1'})
    assert rc == 0, out[-3000:]
    assert 'box coords:' in out and 'labels:' in out
    from PIL import Image
    with Image.open(tmp_path / 'detection_result.jpg') as im:        # cv2.imwrite of the shim
        assert im.size == (1296, 729)


def _tiny_video(path, frames=5, size=(320, 240), fps=12.0, seed=3):
    """A short Motion-JPEG AVI of smooth random frames, written by this package's own writer; returns the frames."""
    from PIL import Image
    from yolov3_tensorflow_amd.utils.video_utils import MjpegAviWriter
    rng = np.random.RandomState(seed)
    made = []
    with MjpegAviWriter(str(path), fps, size, quality=92) as w:
        for _ in range(frames):
            small = rng.randint(0, 256, (size[1] // 16, size[0] // 16, 3)).astype(np.uint8)
            made.append(np.asarray(Image.fromarray(small).resize(size, Image.BICUBIC)))
            w.write(made[-1])
    return made


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is not on this machine')
def test_reference_video_script_runs_unchanged_in_dry_run(tmp_path):
    """video_test.py of the reference, byte-unchanged: cv2.VideoCapture / VideoWriter of the shim over a Motion-JPEG AVI,
    per-frame letterbox, Session.run, box drawing, putText, imshow / waitKey, and the saved result video."""
    os.makedirs(tmp_path / 'data')
    for f in ('yolo_anchors.txt', 'coco.names'):
        with open(os.path.join(ROOT, 'data', f)) as src, open(tmp_path / 'data' / f, 'w') as dst:
            dst.write(src.read())
    _tiny_video(tmp_path / 'in.avi', frames=4)
    rc, out = _run(os.path.join(REF, 'video_test.py'), [str(tmp_path / 'in.avi'), '--save_video', 'true'], str(tmp_path),
                   {'Y3_COMPAT_DRY_RUN': '1'})
    assert rc == 0, out[-3000:]
    assert 'writing Motion-JPEG to video_result.avi' in out           # asked for mp4v into video_result.mp4
    from yolov3_tensorflow_amd.utils.video_utils import open_video
    result = open_video(str(tmp_path / 'video_result.avi'))
    assert (result.frame_count, result.width, result.height, int(result.fps)) == (4, 320, 240, 12)
    assert result.read().shape == (240, 320, 3)


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is not on this machine')
def test_reference_convert_script_runs_unchanged_in_dry_run(tmp_path):
    os.makedirs(tmp_path / 'data' / 'darknet_weights')
    with open(ANCHORS_TXT) as src, open(tmp_path / 'data' / 'yolo_anchors.txt', 'w') as dst:
        dst.write(src.read())
    np.array([0, 2, 0, 0, 0], np.int32).tofile(str(tmp_path / 'data' / 'darknet_weights' / 'yolov3.weights'))
    rc, out = _run(os.path.join(REF, 'convert_weight.py'), [], str(tmp_path), {'Y3_COMPAT_DRY_RUN': '1'})
    assert rc == 0, out[-3000:]
    assert 'checkpoint has been saved' in out


def _tiny_eval_set(root, count=3, seed=0):
    """A few random PNGs + an annotation file in the reference's line format; returns the file's path."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'imgs'), exist_ok=True)
    lines = []
    for i in range(count):
        w, h = int(rng.randint(200, 400)), int(rng.randint(150, 300))
        path = os.path.join(root, 'imgs', '%d.png' % i)
        Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(path)
        lines.append('%d %s %d %d 3 20 30 120 140 17 50 60 150 145' % (i, path, w, h))
    val = os.path.join(root, 'val.txt')
    with open(val, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return val


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is not on this machine')
def test_reference_eval_script_runs_unchanged_in_dry_run(tmp_path):
    """tf.data pipeline + py_func feeder (which really reads and resizes the images), flag placeholders, the per-image
    Session.run loop and the mAP report of the reference's eval.py, byte-unchanged, without a device."""
    os.makedirs(tmp_path / 'data')
    for f in ('yolo_anchors.txt', 'coco.names'):
        with open(os.path.join(ROOT, 'data', f)) as src, open(tmp_path / 'data' / f, 'w') as dst:
            dst.write(src.read())
    val = _tiny_eval_set(str(tmp_path))
    rc, out = _run(os.path.join(REF, 'eval.py'), ['--eval_file', val], str(tmp_path), {'Y3_COMPAT_DRY_RUN': '1'})
    assert rc == 0, out[-3000:]
    assert 'final mAP: 0.0000' in out and 'total_loss: 0.000' in out


def _train_dir(root, train_count=7, val_count=2):
    """The working directory the reference's args.py expects (ref: args.py:10-17): ./data/my_data/{train,val}.txt,
    ./data/yolo_anchors.txt, ./data/coco.names."""
    import shutil
    os.makedirs(os.path.join(root, 'data', 'my_data'))
    for f in ('yolo_anchors.txt', 'coco.names'):
        shutil.copy(os.path.join(ROOT, 'data', f), os.path.join(root, 'data', f))
    ann = _tiny_eval_set(root, count=train_count, seed=3)
    lines = open(ann).read().splitlines()
    with open(os.path.join(root, 'data', 'my_data', 'train.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    with open(os.path.join(root, 'data', 'my_data', 'val.txt'), 'w') as f:
        f.write('\n'.join(lines[:val_count]) + '\n')


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is not on this machine')
def test_reference_train_script_runs_unchanged_in_dry_run(tmp_path):
    """The reference's train.py AND args.py, byte-unchanged, for their full 100 epochs on a 7-image set without a device:
    the re-initialisable iterator over the train / val pipelines (the feeder really reads, mixes up, augments and resizes
    at the multi-scale sizes), the loss / regulariser / summary graph, the warm-up tf.cond around the piecewise
    schedule, compute_gradients -> clip_by_norm -> apply_gradients, the 49 validation passes with the mAP report and the
    checkpoint calls."""
    _train_dir(str(tmp_path))
    rc, out = _run(os.path.join(REF, 'train.py'), [], str(tmp_path), {'Y3_COMPAT_DRY_RUN': '1'})
    assert rc == 0, out[-3000:]
    assert 'start to train' in out
    assert out.count('EVAL: Recall:') == 48            # epochs 4, 6, ..., 98 (ref: train.py:174, args.py:25,68)
    log = open(tmp_path / 'data' / 'progress.log').read()
    assert log.count('======> Epoch:') == 48


def test_tf1_style_training_twin_in_dry_run(tmp_path):
    _train_dir(str(tmp_path))
    out_npz = str(tmp_path / 'steps.npz')
    rc, out = _run(os.path.join(HERE, 'compat_scripts', 'tf1_train.py'),
                   ['--train_file', str(tmp_path / 'data' / 'my_data' / 'train.txt'), '--restore_path', 'none.weights',
                    '--anchor_path', ANCHORS_TXT, '--out', out_npz], str(tmp_path), {'Y3_COMPAT_DRY_RUN': '1'})
    assert rc == 0, out[-3000:]
    d = np.load(out_npz)
    assert d['loss'].shape == (6, 5) and d['lr'].shape == (6,)
    # (no train op runs without a device: the step stays 0 and the warm-up branch of the tf.cond yields 0)
    assert (d['step'] == 0).all() and (d['lr'] == 0).all()


def test_training_graph_symbols():
    """tf.Variable / tf.less / tf.cond / the Node-aware learning-rate schedule / reads-before-updates in one run."""
    from yolov3_tensorflow_amd import compat
    compat.install()
    import tensorflow as tf
    from utils.misc_utils import config_learning_rate
    from yolov3_tensorflow_amd.compat import lazy

    class A(object):
        lr_type, learning_rate_init, pw_boundaries, pw_values = 'piecewise', 1e-2, [5.0], [1e-2, 1e-3]
    step = tf.Variable(0.0, trainable=False, collections=[tf.GraphKeys.LOCAL_VARIABLES])
    rate = tf.cond(tf.less(step, 4), lambda: A.learning_rate_init * step / 4, lambda: config_learning_rate(A, step - 4))
    seen = []

    def bump(r):
        seen.append(float(r))
        step.assign_add(1)
    op = lazy.Node(bump, (rate,), name='train_op')
    op.late = True
    got = []
    with tf.Session() as sess:
        for _ in range(12):
            _, s, r = sess.run([op, step, rate])          # the op is listed first and still runs last
            got.append((float(s), float(r)))
    assert [g[0] for g in got] == [float(i) for i in range(12)]
    want = [1e-2 * i / 4 for i in range(4)] + [1e-2] * 6 + [1e-3] * 2        # piecewise: step - 4 <= 5 -> first value
    assert np.allclose([g[1] for g in got], want) and np.allclose(seen, want)
    # every schedule of utils/misc_utils.py:129-148 as a graph tensor == the eager schedule at the same step
    from yolov3_tensorflow_amd.utils import misc_utils as native

    class S(object):
        learning_rate_init, lr_decay_factor, lr_decay_freq, lr_lower_bound = 1e-3, 0.9, 7, 1e-6
        total_epoches, use_warm_up, warm_up_epoch, train_batch_num = 5, True, 1, 10
        pw_boundaries, pw_values = [12.0, 30.0], [1e-3, 3e-4, 1e-4]
    at = tf.placeholder(tf.float32, [], name='at')
    for kind in ('fixed', 'exponential', 'cosine_decay', 'cosine_decay_restart', 'piecewise'):
        S.lr_type = kind
        node = config_learning_rate(S, at)
        with tf.Session() as sess:
            for g in (0.0, 6.0, 7.0, 13.0, 29.0, 31.0):
                assert np.isclose(float(sess.run(node, feed_dict={at: g})), native.config_learning_rate(S, g), rtol=1e-6), (kind, g)
    # the four optimizers come back as graph-side objects with the reference's call surface
    from utils.misc_utils import config_optimizer
    for name in ('sgd', 'momentum', 'adam', 'rmsprop'):
        opt = config_optimizer(name, rate)
        assert opt.kind == name and hasattr(opt, 'compute_gradients') and hasattr(opt, 'apply_gradients') and hasattr(opt, 'minimize')
    with pytest.raises(ValueError):
        config_optimizer('lamb', rate)
    with pytest.raises(ValueError):
        A.lr_type = 'nope'
        config_learning_rate(A, step)
    with pytest.raises(NotImplementedError):
        tf.data.TextLineDataset(__file__).map(lambda x: x).batch(2)           # wrong order: refused, not mis-run
    assert tf.get_collection(tf.GraphKeys.UPDATE_OPS) == []
    assert -np.Inf == -np.inf                              # NumPy-1 spelling the reference's train.py uses


def test_tf_data_pipeline_and_py_func(tmp_path):
    from yolov3_tensorflow_amd import compat
    from yolov3_tensorflow_amd.compat import lazy
    compat.install()
    try:
        import tensorflow as tf
        p = tmp_path / 'lines.txt'
        p.write_text('a 1\nb 2\nc 3\nd 4\ne 5\n')
        seen = []

        def host(batch, k):
            seen.append([x.decode() for x in batch])
            return np.int64(len(batch) * k), np.asarray([float(x.split()[1]) for x in batch], np.float32)
        ds = tf.data.TextLineDataset(str(p)).batch(2).map(lambda x: tf.py_func(host, [x, 10], [tf.int64, tf.float32]))
        ds.prefetch(3)                                          # result discarded, like ref eval.py:82
        n, v = ds.make_one_shot_iterator().get_next()
        v.set_shape([None])
        with tf.Session() as sess:
            got = [sess.run([n, v]) for _ in range(3)]
            assert [int(g[0]) for g in got] == [20, 20, 10] and got[2][1].tolist() == [5.0]
            assert got[0][0].dtype == np.int64 and got[0][1].dtype == np.float32
            assert seen == [['a 1', 'b 2'], ['c 3', 'd 4'], ['e 5']]      # one pull per Session.run, shared by both outputs
            with pytest.raises(tf.errors.OutOfRangeError):
                sess.run(n)
        assert isinstance(n, lazy.Node) and n.host                    # host-only work also runs in a dry run
    finally:
        sys.path.remove(compat.SHIM_DIR)
        for m in [m for m in sys.modules if m.split('.')[0] in ('tensorflow', 'cv2')]:
            del sys.modules[m]


def test_cv2_shim_known_values(tmp_path):
    from yolov3_tensorflow_amd import compat
    compat.install()
    try:
        import cv2
        assert 'compat' in cv2.__version__
        rgb = np.zeros((4, 6, 3), np.uint8)
        rgb[..., 0], rgb[..., 2] = 200, 50
        from PIL import Image
        Image.fromarray(rgb).save(str(tmp_path / 'a.png'))
        bgr = cv2.imread(str(tmp_path / 'a.png'))
        assert bgr.shape == (4, 6, 3) and (bgr[..., 0] == 50).all() and (bgr[..., 2] == 200).all()     # B,G,R order
        assert cv2.imread(str(tmp_path / 'missing.png')) is None
        assert (cv2.cvtColor(bgr, cv2.COLOR_BGR2RGB) == rgb).all()
        assert cv2.imwrite(str(tmp_path / 'b.png'), bgr)
        assert (np.asarray(Image.open(str(tmp_path / 'b.png'))) == rgb).all()
        assert cv2.resize(bgr, (3, 2)).shape == (2, 3, 3)                                                # dsize = (w, h)
        assert cv2.resize(bgr, (12, 8), interpolation=cv2.INTER_NEAREST).shape == (8, 12, 3)
        canvas = np.zeros((40, 60, 3), np.uint8)
        cv2.rectangle(canvas, (5, 5), (30, 20), [255, 0, 0], thickness=1)                                # blue in BGR
        assert (canvas[5, 10] == [255, 0, 0]).all() and (canvas[10, 10] == 0).all()
        cv2.rectangle(canvas, (40, 15), (50, 5), [0, 0, 255], -1)                                        # corners in any order
        assert (canvas[10, 45] == [0, 0, 255]).all()
        (tw, th), base = cv2.getTextSize('person', 0, fontScale=0.5, thickness=1)
        assert tw > th > 0 and base > 0
        cv2.putText(canvas, 'x', (10, 35), 0, 0.5, [255, 255, 255])
        assert canvas[20:40].any()
        assert cv2.waitKey(0) == -1 and cv2.imshow('w', canvas) is None
    finally:
        sys.path.remove(compat.SHIM_DIR)
        for m in [m for m in sys.modules if m == 'cv2' or m.startswith('cv2.')]:
            del sys.modules[m]


def test_deferred_graph_rules():
    from yolov3_tensorflow_amd.compat import lazy
    calls = []
    x = lazy.Placeholder(np.float32, [1, None, 3], name='x')
    y = lazy.Node(lambda a: (calls.append(1), a * 2)[1], (x,), name='double')
    z = y * y + 1.0
    a, b = lazy.multi(lambda v: (v.sum(), v.shape), (z,), 2, 'stats')
    feed = {x: np.ones((1, 4, 3), np.float32)}
    out = lazy.evaluate([z, a, b, y], feed)
    assert out[0].shape == (1, 4, 3) and (out[0] == 5.0).all() and out[1] == 60.0 and out[2] == (1, 4, 3)
    assert len(calls) == 1                                          # every node is evaluated once per run
    with pytest.raises(ValueError):
        lazy.evaluate(z, {})                                        # unfed placeholder
    with pytest.raises(ValueError):
        lazy.evaluate(z, {x: np.ones((2, 4, 3), np.float32)})       # static dimension mismatch

    class Op(object):
        ran = 0

        def run(self):
            Op.ran += 1
    assert lazy.evaluate([Op(), z], feed)[0] is None and Op.ran == 1
    assert lazy.evaluate([z], feed, dry=True) == [None]


@pytest.mark.gpu
def test_tf1_style_scripts_reproduce_the_golden_detections(tmp_path):
    from oracle import yolo_ref
    weights = str(tmp_path / 'synthetic.weights')
    yolo_ref.write_darknet(yolo_ref.synthetic_params(80, seed=1), weights)
    ckpt = str(tmp_path / 'yolov3.ckpt')
    rc, out = _run(os.path.join(HERE, 'compat_scripts', 'tf1_convert.py'), [weights, ckpt, ANCHORS_TXT], ROOT)
    assert rc == 0 and '366 variables' in out, out[-3000:]
    assert os.path.exists(ckpt + '.npz')
    g = np.load(os.path.join(HERE, 'golden', 'messi_config1_golden.npz'))
    res, jpg = str(tmp_path / 'det.npz'), str(tmp_path / 'det.jpg')
    rc, out = _run(os.path.join(HERE, 'compat_scripts', 'tf1_detect.py'),
                   [os.path.join(HERE, 'golden', 'messi.jpg'), ckpt, ANCHORS_TXT, repr(float(g['score_thresh'])), res, jpg],
                   ROOT)
    assert rc == 0, out[-3000:]
    d = np.load(res)
    assert abs(len(d['labels']) - len(g['labels'])) <= 2, (len(d['labels']), len(g['labels']))
    # match detections by (label, nearest box); near-threshold candidates may differ (see tests/test_messi_gpu.py)
    matched = 0
    for lab, box, sc in zip(g['labels'], g['boxes'], g['scores']):
        cand = np.where(d['labels'] == lab)[0]
        if len(cand) == 0:
            continue
        err = np.abs(d['boxes'][cand] - box).max(axis=1)
        j = cand[err.argmin()]
        scale = max(np.abs(box).max(), 1.0)
        if err.min() <= 1e-3 * scale + 1e-3 and abs(d['scores'][j] - sc) <= 1e-3:
            matched += 1
    assert matched >= len(g['labels']) - 2, (matched, len(g['labels']))
    from PIL import Image
    with Image.open(jpg) as im:
        assert im.size == (1296, 729)


@pytest.mark.gpu
def test_tf1_style_eval_matches_the_native_eval_script(tmp_path):
    """The TF-1 style evaluation twin (tf.data + py_func feeder, flag placeholders, per-image Session.run) against this
    package's own eval.py on the same tiny annotation file and weights: same mAP, same detections, same mean loss."""
    import json
    from oracle import yolo_ref
    weights = str(tmp_path / 'synthetic.weights')
    yolo_ref.write_darknet(yolo_ref.synthetic_params(80, seed=1), weights)
    val = _tiny_eval_set(str(tmp_path), count=4, seed=3)
    rep = str(tmp_path / 'report.json')
    rc, out = _run(os.path.join(HERE, 'compat_scripts', 'tf1_eval.py'),
                   ['--eval_file', val, '--restore_path', weights, '--anchor_path', ANCHORS_TXT, '--json', rep,
                    '--score_threshold', '0.02'], ROOT)
    assert rc == 0, out[-3000:]
    got = json.load(open(rep))
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'eval.py'), '--eval_file', val, '--restore_path', weights,
                        '--anchor_path', ANCHORS_TXT, '--class_name_path', os.path.join(ROOT, 'data', 'coco.names'),
                        '--batch_size', '1', '--score_threshold', '0.02'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    native = r.stdout.decode(errors='replace')
    assert r.returncode == 0, native[-3000:]
    import re
    m_ap = float(re.search(r'final mAP: ([0-9.naNA]+)', native).group(1))        # nan when a class has no ground truth
    total = float(re.search(r'total_loss: ([0-9.]+)', native).group(1))        # (0/0 in voc_eval, like the reference)
    assert np.isclose(got['mAP'], m_ap, atol=1e-4, equal_nan=True), (got, m_ap)
    per_class = dict((int(c), float(a)) for c, a in re.findall(r'Class (\d+): .*AP: ([0-9.]+)', native))
    assert per_class, native[-2000:]
    assert abs(got['loss'][0] - total) <= 1e-3 * max(total, 1.0) + 6e-4, (got, total)      # the report prints 3 decimals
    assert got['detections'] > 0


@pytest.mark.gpu
def test_tf1_style_training_twin_matches_the_native_trainer(tmp_path, isolated_graph):
    """Six training steps through the compat training graph (tf.data feeder -> forward(is_training) -> loss ->
    compute_gradients / clip_by_norm / apply_gradients with a warm-up tf.cond) against the same six steps through
    yolov3_tensorflow_amd.training.Trainer on the same batches and weights: same losses, same learning rates, same
    updated variables and moving statistics (the two run the same library calls: tolerance = atomics' ordering)."""
    import torch
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils import misc_utils, data_utils
    from oracle import yolo_ref
    params = yolo_ref.synthetic_params(80, seed=1)
    weights = str(tmp_path / 'synthetic.weights')
    yolo_ref.write_darknet(params, weights)
    _train_dir(str(tmp_path), train_count=8)
    train_file = str(tmp_path / 'data' / 'my_data' / 'train.txt')
    out_npz = str(tmp_path / 'steps.npz')
    rc, out = _run(os.path.join(HERE, 'compat_scripts', 'tf1_train.py'),
                   ['--train_file', train_file, '--restore_path', weights, '--anchor_path', ANCHORS_TXT, '--out', out_npz],
                   str(tmp_path))
    assert rc == 0, out[-3000:]
    got = np.load(out_npz)
    assert list(got['step']) == [0., 1., 2., 3., 4., 5.]
    want_lr = [0.0, 5e-4, 1e-3, 1e-3, 1e-3, 1e-3]          # warm-up over 2 steps, then piecewise (boundary at 4 + 2)
    assert np.allclose(got['lr'], want_lr)
    assert list(got['boxes_shape']) == [4, 3 * (8 * 8 + 16 * 16 + 32 * 32), 4]

    # the native replay
    y3.reset_default_graph()
    anchors = misc_utils.parse_anchors(ANCHORS_TXT)
    model = y3.yolov3(80, anchors, True, True, 0.9, 5e-4, use_static_shape=False)
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 32, 32, 3))
    misc_utils.run_ops(misc_utils.load_weights(y3.global_variables(scope='yolov3'), weights))
    upd = [v for v in y3.global_variables(scope='yolov3') if v.op_name.startswith('yolov3/yolov3_head/')]
    opt = training.Optimizer('momentum', 0.0, momentum=0.9)
    trainer = training.Trainer(model, opt, update_vars=upd, clip_norm=100.)
    lines = open(train_file).read().splitlines()
    batches = [lines[0:4], lines[4:8]]
    losses = []
    for i in range(6):
        _, img, y13, y26, y52 = data_utils.get_batch_data(np.asarray(batches[i % 2], dtype=object), 80, [256, 256], anchors,
                                                          'val', False, False, False)
        opt.learning_rate = want_lr[i]
        with y3.variable_scope('yolov3'):
            losses.append([float(v) for v in trainer.step(img, [y13, y26, y52])])
    np.testing.assert_allclose(got['loss'], np.array(losses), rtol=2e-5, atol=1e-6)
    by_name = dict((v.op_name, v) for v in y3.global_variables(scope='yolov3'))
    for key in got.files:
        if not key.startswith('yolov3.'):
            continue
        mine = by_name[key.replace('.', '/')].tensor.cpu().numpy()
        np.testing.assert_allclose(got[key], mine, rtol=2e-5, atol=1e-6, err_msg=key)
    # the body was not in update_part: its kernel still equals the checkpoint's, its moving statistics moved
    assert np.array_equal(got['yolov3.darknet53_body.Conv.weights'], params['yolov3/darknet53_body/Conv/weights'])
    assert not np.allclose(got['yolov3.darknet53_body.Conv.BatchNorm.moving_variance'],
                           params['yolov3/darknet53_body/Conv/BatchNorm/moving_variance'])
    assert got['reg'][0] > 0 and np.isfinite(got['val_loss']).all()
