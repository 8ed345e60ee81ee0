"""SURVEY §8(f) row 2 — evaluation bookkeeping against golden vectors produced by the reference's own numpy
functions (tests/golden/make_golden.py::eval_goldens imports utils/eval_utils.py unmodified).  Everything here is
float64 host arithmetic: the bar is bit-exact.  The CPU tests use the oracle's python NMS as the NMS stand-in;
the GPU tests run evaluate_on_cpu / evaluate_on_gpu / get_preds through the HIP NMS kernels."""
import functools
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(HERE, 'golden', 'reference_eval_goldens.npz'))


def _gt_from_flat(flat):
    d = {}
    for row in flat:
        d.setdefault(int(row[0]), []).append([row[1], row[2], row[3], row[4], int(row[5])])
    return d


def test_calc_iou_bit_exact(g):
    from yolov3_tensorflow_amd.utils import eval_utils
    np.testing.assert_array_equal(eval_utils.calc_iou(g['iou_pred'], g['iou_true']), g['iou_out'])
    assert eval_utils.calc_iou(np.zeros((0, 4)), g['iou_true']).shape == (0, 11)


def test_voc_ap_both_metrics_bit_exact(g):
    from yolov3_tensorflow_amd.utils import eval_utils
    assert eval_utils.voc_ap(g['ap_rec'], g['ap_prec'], False) == float(g['ap_area'])
    assert eval_utils.voc_ap(g['ap_rec'], g['ap_prec'], True) == float(g['ap_07'])
    # perfect detector: AP = 1 under both metrics
    rec = np.arange(1, 11) / 10.0
    assert eval_utils.voc_ap(rec, np.ones(10)) == 1.0
    assert abs(eval_utils.voc_ap(rec, np.ones(10), True) - 1.0) < 1e-12


def test_parse_line_and_errors(g):
    from yolov3_tensorflow_amd.utils import data_utils
    line = str(g['ann_lines'][3])
    for ln in (line, line.encode()):
        idx, path, boxes, labels, w, h = data_utils.parse_line(ln)
        assert (idx, path) == (int(g['pl_idx']), str(g['pl_path']))
        assert boxes.dtype == np.float32 and labels.dtype == np.int64
        np.testing.assert_array_equal(boxes, g['pl_boxes'])
        np.testing.assert_array_equal(labels, g['pl_labels'])
        assert [w, h] == g['pl_wh'].tolist()
    with pytest.raises(AssertionError):
        data_utils.parse_line('0 img.jpg 640 480')                     # no object
    with pytest.raises(AssertionError):
        data_utils.parse_line('0 img.jpg 640 480 1 2.0 3.0 4.0 5.0 7')   # ragged object record


@pytest.mark.parametrize('letterbox', [True, False])
def test_parse_gt_rec_bit_exact(g, tmp_path, letterbox):
    from yolov3_tensorflow_amd.utils import eval_utils
    ann = tmp_path / 'val.txt'
    ann.write_text('\n'.join(str(l) for l in g['ann_lines']) + '\n')
    eval_utils.gt_dict = {}
    gd = eval_utils.parse_gt_rec(str(ann), [416, 416], letterbox)
    flat = np.array([[k] + [float(v) for v in o] for k, v in sorted(gd.items()) for o in v], np.float64)
    np.testing.assert_array_equal(flat, g['gt_rec_lb%d' % int(letterbox)])
    assert eval_utils.parse_gt_rec('/nonexistent', [1, 1]) is gd        # module-level cache, like the reference
    eval_utils.gt_dict = {}


def test_voc_eval_bit_exact_per_class(g, capsys):
    from yolov3_tensorflow_amd.utils import eval_utils
    gd = _gt_from_flat(g['gt_rec_lb1'])
    preds = [[int(r[0]), r[1], r[2], r[3], r[4], r[5], int(r[6])] for r in g['voc_preds']]
    for row in g['voc_results']:
        cls, m07 = int(row[0]), bool(row[1])
        got = eval_utils.voc_eval(gd, preds, cls, iou_thres=0.5, use_07_metric=m07)
        assert [float(v) for v in got] == row[2:].tolist(), (cls, m07)
    assert 'no box, ignore' in capsys.readouterr().out                  # the class with no detection


def _numpy_cpu_nms(num_classes, max_boxes, score_thresh, iou_thresh, boxes, scores):
    from oracle import nms_ref
    b, s, l, _ = nms_ref.per_class('py', boxes, scores, num_classes, max_boxes, score_thresh, iou_thresh)
    return (None, None, None) if len(l) == 0 else (b, s, l)


def test_evaluate_counts_with_oracle_nms(g):
    """_evaluate (the shared body of evaluate_on_cpu/gpu) with the oracle's numpy cpu_nms as the NMS."""
    from yolov3_tensorflow_amd.utils import eval_utils
    C = g['ev_dicts'].shape[0]
    y_pred = [g['ev_pred_boxes'], g['ev_pred_confs'], g['ev_pred_probs']]
    y_true = [g['ev_ytrue%d' % j] for j in range(3)]
    nms = functools.partial(_numpy_cpu_nms, C, 50, 0.3, 0.5)
    rec, prec = eval_utils.evaluate_on_gpu(None, nms, None, None, y_pred, y_true, C, iou_thresh=0.5, calc_now=True)
    assert [rec, prec] == g['ev_recall_precision'].tolist()
    tp, tl, pl = eval_utils.evaluate_on_gpu(None, nms, None, None, y_pred, y_true, C, iou_thresh=0.5, calc_now=False)
    got = np.array([[tp[c], tl[c], pl[c]] for c in range(C)], np.int64)
    np.testing.assert_array_equal(got, g['ev_dicts'])


@pytest.mark.gpu
def test_evaluate_on_cpu_and_gpu_through_the_hip_nms(g):
    import torch
    from yolov3_tensorflow_amd.utils import eval_utils, nms_utils
    C = g['ev_dicts'].shape[0]
    y_pred = [g['ev_pred_boxes'], g['ev_pred_confs'], g['ev_pred_probs']]
    y_true = [g['ev_ytrue%d' % j] for j in range(3)]
    rec, prec = eval_utils.evaluate_on_cpu(y_pred, y_true, C, calc_now=True, max_boxes=50, score_thresh=0.3,
                                           iou_thresh=0.5)
    assert [rec, prec] == g['ev_recall_precision'].tolist()
    tp, tl, pl = eval_utils.evaluate_on_cpu(y_pred, y_true, C, calc_now=False, max_boxes=50, score_thresh=0.3,
                                            iou_thresh=0.5)
    np.testing.assert_array_equal(np.array([[tp[c], tl[c], pl[c]] for c in range(C)]), g['ev_dicts'])
    # the gpu_nms op (TF semantics) on device tensors: same ground-truth counts, sane recall
    op = functools.partial(nms_utils.gpu_nms, num_classes=C, max_boxes=50, score_thresh=0.3, nms_thresh=0.5)
    yp = [torch.from_numpy(a).cuda() for a in y_pred]
    tp2, tl2, pl2 = eval_utils.evaluate_on_gpu(None, op, None, None, yp, y_true, C, iou_thresh=0.5, calc_now=False)
    assert tl2 == tl and sum(tp2.values()) >= sum(tp.values()) - 2
    # get_preds_gpu rows == get_preds_batch rows for the same detections
    rows1 = eval_utils.get_preds_gpu(None, op, None, None, [7], [a[1:2] for a in yp])
    b, s = yp[0][1:2], (yp[1] * yp[2])[1:2]
    det = nms_utils.gpu_nms_batched(b, s, C, 50, 0.3, 0.5)
    rows2 = eval_utils.get_preds_batch([7], det)
    assert len(rows1) == len(rows2) > 0
    for r1, r2 in zip(rows1, rows2):
        assert r1[0] == r2[0] == 7 and [float(v) for v in r1[1:]] == [float(v) for v in r2[1:]]


@pytest.mark.gpu
def test_eval_script_batched_equals_per_image(tmp_path, capsys, isolated_graph):
    """eval.py end to end on a synthetic 6-image / 5-class set with random weights: the batched device path
    (--batch_size 4, ragged last batch) reports what the one-image-per-step path (--batch_size 1, the reference's
    shape) reports; the report has the reference's lines."""
    import sys
    from PIL import Image
    import torch
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.utils import misc_utils
    sys.path.insert(0, os.path.dirname(HERE))
    import eval as eval_script
    rng = np.random.RandomState(5)
    names = tmp_path / 'names.txt'
    names.write_text('\n'.join('c%d' % i for i in range(5)) + '\n')
    lines = []
    obj = 0          # labels go round-robin: every class has ground truth (a class without any makes the
    for i in range(6):  # reference's recall 0/0 = NaN, and ours likewise)
        w, h = int(rng.randint(120, 400)), int(rng.randint(120, 400))
        img = (rng.rand(h // 8 + 1, w // 8 + 1, 3) * 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w]
        path = tmp_path / ('im%d.png' % i)
        Image.fromarray(img).save(str(path))
        parts = ['%d' % i, str(path), '%d' % w, '%d' % h]
        for _ in range(int(rng.randint(2, 4))):
            x0, y0 = rng.uniform(0, w * 0.6), rng.uniform(0, h * 0.6)
            obj += 1
            parts += ['%d' % (obj % 5), '%.1f' % x0, '%.1f' % y0, '%.1f' % (x0 + rng.uniform(20, w * 0.35)),
                      '%.1f' % (y0 + rng.uniform(20, h * 0.35))]
        lines.append(' '.join(parts))
    ann = tmp_path / 'val.txt'
    ann.write_text('\n'.join(lines) + '\n')
    anchors = os.path.join(os.path.dirname(HERE), 'data', 'yolo_anchors.txt')
    # random weights with a small detection-conv gain so that logits stay O(1)
    y3.reset_default_graph()
    y3.set_init_seed(3)
    m = y3.yolov3(5, misc_utils.parse_anchors(anchors))
    with y3.variable_scope('yolov3'):
        m.forward(torch.zeros(1, 64, 64, 3))
    wfile = str(tmp_path / 'rand.weights')
    misc_utils.save_weights(y3.global_variables(scope='yolov3'), wfile)
    results = {}
    for bs in (4, 1):
        y3.reset_default_graph()
        results[bs] = eval_script.main(['--eval_file', str(ann), '--restore_path', wfile, '--anchor_path', anchors,
                                        '--class_name_path', str(names), '--img_size', '224', '160',
                                        '--letterbox_resize', 'true', '--batch_size', str(bs),
                                        '--score_threshold', '0.1', '--nms_topk', '20'])
    out = capsys.readouterr().out
    assert 'final mAP:' in out and 'Class 4: Recall:' in out and 'total_loss:' in out
    a, b = results[4], results[1]
    assert 0.0 <= a['mAP'] <= 1.0 and np.isfinite(a['loss']).all()
    np.testing.assert_allclose(a['loss'], b['loss'], rtol=1e-4)
    # the K sum of a stream-K conv is associated differently at different batch sizes, so a
    # detection sitting exactly on an NMS threshold may flip: compare the sets up to a handful of rows
    key = lambda r: (r[0], int(r[6]), round(float(r[5]), 4), round(float(r[1]), 1))
    sa, sb = set(map(key, a['val_preds'])), set(map(key, b['val_preds']))
    assert len(sa) > 50 and len(sa ^ sb) <= 0.02 * len(sa), (len(sa), len(sa ^ sb))
    assert abs(a['mAP'] - b['mAP']) < 0.02


def test_color_table_matches_the_reference(g):
    from yolov3_tensorflow_amd.utils.plot_utils import get_color_table, plot_one_box
    table = get_color_table(80)
    np.testing.assert_array_equal(np.array([table[i] for i in range(80)]), g['color_table_80'])
    img = np.zeros((120, 160, 3), np.uint8)
    plot_one_box(img, [20, 40, 100, 90], label='cat, 97.00%', color=table[3])
    assert (img[40, 20:101] == table[3]).all() and (img[65, 60] == 0).all()      # outline drawn, interior untouched
    assert (img[30:40, 20:60] != 0).any()                                         # caption strip above the box


@pytest.mark.gpu
def test_single_image_script(tmp_path, capsys, isolated_graph):
    """test_single_image.py twin on a synthetic picture with random weights: both resize modes run, print the three
    blocks the reference prints, write the annotated file, and return detections inside the picture frame."""
    import sys
    from PIL import Image
    import yolov3_tensorflow_amd as y3
    sys.path.insert(0, os.path.dirname(HERE))
    import test_single_image as script
    rng = np.random.RandomState(9)
    pic = (rng.rand(30, 44, 3) * 255).astype(np.uint8).repeat(10, 0).repeat(10, 1)     # 300 x 440
    src = tmp_path / 'pic.png'
    Image.fromarray(pic).save(str(src))
    root = os.path.dirname(HERE)
    for letterbox in ('true', 'false'):
        y3.reset_default_graph()
        y3.set_init_seed(11)
        out = tmp_path / ('out_%s.png' % letterbox)
        boxes, scores, labels = script.main([str(src), '--anchor_path', os.path.join(root, 'data', 'yolo_anchors.txt'),
                                             '--class_name_path', os.path.join(root, 'data', 'coco.names'),
                                             '--new_size', '320', '224', '--letterbox_resize', letterbox,
                                             '--restore_path', str(tmp_path / 'missing.weights'), '--output', str(out)])
        text = capsys.readouterr().out
        assert 'box coords:' in text and 'scores:' in text and 'labels:' in text and 'not found' in text
        assert boxes.shape[1] == 4 and len(boxes) == len(scores) == len(labels)
        assert (scores >= 0.3).all() and labels.dtype == np.int32
        assert Image.open(str(out)).size == (440, 300)
