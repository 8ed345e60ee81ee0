"""The reference's OWN driver scripts (convert_weight.py, test_single_image.py, video_test.py, eval.py, train.py + args.py), byte-unchanged,
run for real on the device through `python -m yolov3_tensorflow_amd.compat.run`.

The scripts are not part of this repository and /root/reference does not exist on the GPU box: tools/gpu_reference_scripts.sh
stages them (unmodified) under the git-ignored oracle/_ref/scripts/ for one gpurun call and removes them afterwards;
Y3_REFERENCE_SCRIPTS may name another directory holding them.  Without them every test here is skipped - the always-on
coverage of the same symbols is tests/test_compat.py (TF-1 style twins in tests/compat_scripts)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SCRIPTS = os.environ.get('Y3_REFERENCE_SCRIPTS') or os.path.join(ROOT, 'oracle', '_ref', 'scripts')
NEEDED = ('convert_weight.py', 'test_single_image.py', 'video_test.py', 'eval.py', 'train.py', 'args.py')

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not all(os.path.exists(os.path.join(SCRIPTS, f)) for f in NEEDED),
                                 reason='the reference scripts are not staged on this machine')]


def _run(script, args, cwd, keep=None, timeout=1500):
    """Run one reference script in `cwd`; `keep` = (npz path, [module-level names]) saved after the script ends."""
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    prog = ("import sys, numpy as np\n"
            "from yolov3_tensorflow_amd.compat.run import run_script\n"
            "g = run_script(sys.argv[1:])\n")
    if keep:
        prog += "np.savez(%r, **dict((k, np.asarray(g[k])) for k in %r))\n" % (keep[0], list(keep[1]))
    r = subprocess.run([sys.executable, '-c', prog, os.path.join(SCRIPTS, script)] + list(args), cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return r.returncode, r.stdout.decode(errors='replace')


@pytest.fixture(scope='module')
def workdir(tmp_path_factory):
    """./data/{yolo_anchors.txt, coco.names, darknet_weights/yolov3.weights, my_data/{train,val}.txt} as the scripts'
    defaults expect them, then the reference's convert_weight.py -> ./data/darknet_weights/yolov3.ckpt."""
    sys.path.insert(0, HERE)
    from test_compat import _train_dir
    from oracle import yolo_ref
    root = str(tmp_path_factory.mktemp('refscripts'))
    _train_dir(root, train_count=7, val_count=3)
    os.makedirs(os.path.join(root, 'data', 'darknet_weights'))
    yolo_ref.write_darknet(yolo_ref.synthetic_params(80, seed=1), os.path.join(root, 'data', 'darknet_weights', 'yolov3.weights'))
    rc, out = _run('convert_weight.py', [], root)
    assert rc == 0 and 'checkpoint has been saved' in out, out[-3000:]
    assert os.path.exists(os.path.join(root, 'data', 'darknet_weights', 'yolov3.ckpt.npz'))
    return root


def test_reference_demo_script_reproduces_the_golden_detections(workdir):
    """convert_weight.py's checkpoint -> test_single_image.py on the demo image (letterbox, score 0.3, NMS 0.45): every
    golden detection (oracle, score threshold 0.367 > 0.3) is among the script's detections, same box and score."""
    res = os.path.join(workdir, 'det.npz')
    rc, out = _run('test_single_image.py', [os.path.join(HERE, 'golden', 'messi.jpg')], workdir,
                   keep=(res, ['boxes_', 'scores_', 'labels_']))
    assert rc == 0, out[-3000:]
    assert 'box coords:' in out and 'scores:' in out and 'labels:' in out
    d, g = np.load(res), np.load(os.path.join(HERE, 'golden', 'messi_config1_golden.npz'))
    assert len(d['labels_']) >= len(g['labels']) - 2
    matched = 0
    for lab, box, sc in zip(g['labels'], g['boxes'], g['scores']):
        cand = np.where(d['labels_'] == lab)[0]
        if len(cand) == 0:
            continue
        err = np.abs(d['boxes_'][cand] - box).max(axis=1)
        j = cand[err.argmin()]
        if err.min() <= 1e-3 * max(np.abs(box).max(), 1.0) + 1e-3 and abs(d['scores_'][j] - sc) <= 1e-3:
            matched += 1
    print('test_single_image.py: %d detections at 0.3; %d of the %d golden ones (score >= %.3f) matched'
          % (len(d['labels_']), matched, len(g['labels']), float(g['score_thresh'])))
    assert matched >= len(g['labels']) - 2, (matched, len(g['labels']))
    from PIL import Image
    with Image.open(os.path.join(workdir, 'detection_result.jpg')) as im:
        assert im.size == (1296, 729)


def test_reference_video_script_detects_on_every_frame(workdir):
    """video_test.py on a three-frame Motion-JPEG AVI of the demo image (the third frame mirrored): the last frame's
    detections are left in the script's globals, the annotated result video has the three frames at the source size."""
    from PIL import Image
    from yolov3_tensorflow_amd.utils.video_utils import MjpegAviWriter, open_video
    picture = np.asarray(Image.open(os.path.join(HERE, 'golden', 'messi.jpg')).convert('RGB').resize((648, 364), Image.BICUBIC))
    clip = os.path.join(workdir, 'clip.avi')
    with MjpegAviWriter(clip, 10, (648, 364), quality=95) as w:
        for frame in (picture, picture, picture[:, ::-1]):
            w.write(frame)
    res = os.path.join(workdir, 'video_det.npz')
    rc, out = _run('video_test.py', [clip, '--save_video', 'true'], workdir, keep=(res, ['boxes_', 'scores_', 'labels_', 'i']))
    assert rc == 0, out[-3000:]
    d = np.load(res)
    print('video_test.py (reference, compat): %d detections on the last of 3 frames, scores %.3f..%.3f'
          % (len(d['labels_']), d['scores_'].min(), d['scores_'].max()))
    assert len(d['labels_']) > 20 and d['scores_'].min() >= 0.3
    assert (d['boxes_'][:, 2] > d['boxes_'][:, 0]).all() and np.isfinite(d['boxes_']).all()
    # this package's own video_test.py on the same clip and weights finds the same number of objects on that frame
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'video_test.py'), clip, '--restore_path',
                        os.path.join(workdir, 'data', 'darknet_weights', 'yolov3.weights'), '--anchor_path',
                        os.path.join(ROOT, 'data', 'yolo_anchors.txt'), '--class_name_path',
                        os.path.join(ROOT, 'data', 'coco.names'), '--batch_size', '1'],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    native = r.stdout.decode(errors='replace')
    assert r.returncode == 0, native[-3000:]
    counts = [int(v) for v in re.findall(r'frame \d+: (\d+) detections', native)]
    print('video_test.py (native twin): detections per frame %s' % counts)
    assert len(counts) == 3 and abs(counts[2] - len(d['labels_'])) <= 2
    result = open_video(os.path.join(workdir, 'video_result.avi'))
    assert (result.frame_count, result.width, result.height) == (3, 648, 364)
    first = result.read()
    assert np.abs(first.astype(int) - picture.astype(int)).mean() > 0.5           # boxes and the time were drawn on it


def test_reference_eval_script_matches_the_native_eval(workdir):
    val = os.path.join(workdir, 'data', 'my_data', 'val.txt')
    rc, out = _run('eval.py', ['--score_threshold', '0.02'], workdir)
    assert rc == 0, out[-3000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'eval.py'), '--eval_file', val, '--restore_path',
                        os.path.join(workdir, 'data', 'darknet_weights', 'yolov3.weights'), '--anchor_path',
                        os.path.join(ROOT, 'data', 'yolo_anchors.txt'), '--class_name_path',
                        os.path.join(ROOT, 'data', 'coco.names'), '--batch_size', '1', '--score_threshold', '0.02'],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    native = r.stdout.decode(errors='replace')
    assert r.returncode == 0, native[-3000:]
    pick = lambda text, pat: [float(v) for v in re.search(pat, text).groups()]
    m1, m2 = pick(out, r'final mAP: ([0-9.na]+)'), pick(native, r'final mAP: ([0-9.na]+)')
    l1 = pick(out, r'total_loss: ([0-9.]+), loss_xy: ([0-9.]+), loss_wh: ([0-9.]+), loss_conf: ([0-9.]+), loss_class: ([0-9.]+)')
    l2 = pick(native, r'total_loss: ([0-9.]+), loss_xy: ([0-9.]+), loss_wh: ([0-9.]+), loss_conf: ([0-9.]+), loss_class: ([0-9.]+)')
    print('eval.py (reference, compat): mAP %s loss %s; native eval.py: mAP %s loss %s' % (m1, l1, m2, l2))
    assert np.allclose(m1, m2, atol=1e-4, equal_nan=True)
    assert np.allclose(l1, l2, rtol=1e-3, atol=2e-3)
    assert re.findall(r'Class (\d+): Recall: ([0-9.]+), Precision: ([0-9.]+), AP: ([0-9.]+)', out) == \
        re.findall(r'Class (\d+): Recall: ([0-9.]+), Precision: ([0-9.]+), AP: ([0-9.]+)', native)


def test_reference_train_script_trains(workdir):
    """train.py + args.py unchanged: 100 epochs x 2 batches (6 + 1 images, multi-scale 320..640, mix-up, augmentation,
    label smoothing, focal loss, momentum on the head with a 3-epoch warm-up), validation + mAP every 2nd epoch from 4,
    checkpoints through tf.train.Saver.  The loss must stay finite and fall."""
    rc, out = _run('train.py', [], workdir, timeout=2400)
    assert rc == 0, out[-4000:]
    log = open(os.path.join(workdir, 'data', 'progress.log')).read()
    evals = re.findall(r'======> Epoch: (\d+), global_step: ([0-9.]+), lr: ([0-9.e-]+) <======', log)
    losses = [float(v) for v in re.findall(r'EVAL: loss: total: ([0-9.naninf]+),', log)]
    train_rows = re.findall(r'Epoch: (\d+), global_step: (\d+) \| loss: total: ([0-9.]+).*rec: ([0-9.]+), prec: ([0-9.]+) \| lr: ([0-9.e-]+)', log)
    print('train.py (reference, compat): %d validation passes; validation loss first %.2f, min %.2f, last %.2f; '
          'train rows %s' % (len(evals), losses[0], min(losses), losses[-1], train_rows))
    assert len(evals) == 48 and len(losses) == 48
    # the step value a run fetches is the one BEFORE its own train op (reads first): last run of epochs 4 and 98
    assert float(evals[0][1]) == 2 * 5 - 1 and float(evals[-1][1]) == 2 * 99 - 1
    assert float(evals[0][2]) == 1e-4                                          # past the 6-step warm-up: pw_values[0]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    assert [int(r[1]) for r in train_rows] == [100]                            # train_evaluation_step (global_step 200 is the 201st run)
    # (no checkpoint is expected: classes without ground truth make the script's mean AP nan, and loss > 2 - ref:
    # train.py:170,213)
    print('checkpoint dir:', os.listdir(os.path.join(workdir, 'checkpoint')) if os.path.isdir(os.path.join(workdir, 'checkpoint')) else None)
