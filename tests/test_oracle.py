"""CPU tests: the oracle against the golden vectors generated from the reference's own numpy functions,
and the oracle's internal consistency (python vs C, fp32 vs fp64)."""
import os
import tempfile

import numpy as np
import pytest

from oracle import nms_ref, yolo_ref
from conftest import make_boxes


def test_py_nms_matches_reference_goldens(golden):
    n = int(golden['py_nms_n'])
    assert n >= 5
    for i in range(n):
        g = {k: golden['py_nms_%d_%s' % (i, k)] for k in ('boxes', 'scores', 'max_boxes', 'iou_thresh', 'keep')}
        want = g['keep'].tolist()
        got_py = nms_ref.py_nms(g['boxes'], g['scores'], int(g['max_boxes']), float(g['iou_thresh']))
        got_c = nms_ref.c_single('py', g['boxes'], g['scores'], int(g['max_boxes']), float(g['iou_thresh']))
        assert got_py == want, 'case %d (python restatement)' % i
        assert got_c == want, 'case %d (C restatement)' % i


def test_cpu_nms_matches_reference_goldens(golden):
    n = int(golden['cpu_nms_n'])
    for i in range(n):
        g = {k: golden['cpu_nms_%d_%s' % (i, k)] for k in
             ('boxes', 'scores', 'num_classes', 'max_boxes', 'score_thresh', 'iou_thresh', 'is_none',
              'out_boxes', 'out_scores', 'out_labels')}
        for fn in (nms_ref.per_class, nms_ref.c_per_class):
            b, s, l, _ = fn('py', g['boxes'], g['scores'], int(g['num_classes']), int(g['max_boxes']),
                            float(g['score_thresh']), float(g['iou_thresh']))
            assert (len(b) == 0) == bool(g['is_none'])
            np.testing.assert_array_equal(b, g['out_boxes'])
            np.testing.assert_array_equal(s, g['out_scores'])
            np.testing.assert_array_equal(l, g['out_labels'])


@pytest.mark.parametrize('mode', ['tf', 'py'])
def test_python_and_c_nms_agree(mode):
    rng = np.random.RandomState(7)
    for trial in range(4):
        B, C = 400, 3
        boxes = make_boxes(rng, B)
        scores = (rng.rand(B, C) * rng.rand(B, C)).astype(np.float32)
        if trial == 1:   # ties
            scores[:200] = np.round(scores[:200] * 8) / 8
        if trial == 2:   # degenerate / inverted boxes
            boxes[:50, 2:] = boxes[:50, :2]
            boxes[50:80] = boxes[50:80][:, [2, 3, 0, 1]]
        a = nms_ref.per_class(mode, boxes, scores, C, 40, 0.1, 0.45)
        b = nms_ref.c_per_class(mode, boxes, scores, C, 40, 0.1, 0.45)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_tf_nms_semantics_small():
    # two identical boxes + one disjoint: IoU(0,1) = 1 > thr -> second suppressed; disjoint kept
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30]], np.float32)
    scores = np.array([0.9, 0.8, 0.7], np.float32)
    assert nms_ref.tf_nms(boxes, scores, 10, 0.5) == [0, 2]
    # IoU exactly equal to the threshold is NOT suppressed (strict >)
    boxes = np.array([[0, 0, 2, 2], [0, 1, 2, 3]], np.float32)   # inter 2, union 6 -> 1/3
    thr = float(np.float32(2.0) / np.float32(6.0))
    assert nms_ref.tf_nms(boxes, np.array([0.9, 0.8], np.float32), 10, thr) == [0, 1]
    # equal scores: lower index first
    assert nms_ref.tf_nms(boxes, np.array([0.5, 0.5], np.float32), 1, 0.9) == [0]
    # max_output_size respected
    assert nms_ref.tf_nms(boxes, np.array([0.1, 0.5], np.float32), 1, 0.9) == [1]
    # empty input
    assert nms_ref.tf_nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 5, 0.5) == []


def test_parse_anchors_golden(golden, anchors):
    np.testing.assert_array_equal(golden['anchors'], anchors)
    assert golden['anchors'].dtype == np.float32


def test_variable_specs_match_survey():
    specs = yolo_ref.variable_specs(80)
    assert len(specs) == 366                                   # 72*5 + 3*2
    assert sum(int(np.prod(s)) for _, s in specs) == 62001757  # floats in yolov3.weights
    names = [n for n, _ in specs]
    assert names[0] == 'yolov3/darknet53_body/Conv/weights'
    assert names[1] == 'yolov3/darknet53_body/Conv/BatchNorm/gamma'
    assert names[2] == 'yolov3/darknet53_body/Conv/BatchNorm/beta'
    assert 'yolov3/yolov3_head/Conv_6/biases' in names and 'yolov3/yolov3_head/Conv_22/biases' in names
    assert dict(specs)['yolov3/yolov3_head/Conv_8/weights'] == (1, 1, 768, 256)
    assert dict(specs)['yolov3/yolov3_head/Conv_16/weights'] == (1, 1, 384, 128)


def test_darknet_file_roundtrip_and_size():
    params = yolo_ref.synthetic_params(80, seed=3)
    path = os.path.join(tempfile.mkdtemp(), 'w.weights')
    yolo_ref.write_darknet(params, path)
    assert os.path.getsize(path) == 248007048
    back = yolo_ref.read_darknet(path)
    assert list(back.keys()) == list(params.keys())
    for k in params:
        np.testing.assert_array_equal(params[k], back[k])
    # file order inside a BN'd conv: beta, gamma, mean, var, then OIHW kernel
    raw = np.fromfile(path, np.float32, offset=20, count=32 * 4 + 864)
    np.testing.assert_array_equal(raw[:32], params['yolov3/darknet53_body/Conv/BatchNorm/beta'])
    np.testing.assert_array_equal(raw[32:64], params['yolov3/darknet53_body/Conv/BatchNorm/gamma'])
    w = params['yolov3/darknet53_body/Conv/weights']
    np.testing.assert_array_equal(raw[128:], np.transpose(w, (3, 2, 0, 1)).ravel())


def test_forward_fp32_close_to_fp64_and_shapes():
    import torch
    params = yolo_ref.synthetic_params(80, seed=1)
    x = np.random.RandomState(0).rand(1, 96, 128, 3).astype(np.float32)
    f32 = yolo_ref.forward(params, x)
    f64 = yolo_ref.forward(params, x, dtype=torch.float64)
    assert [f.shape for f in f32] == [(1, 3, 4, 255), (1, 6, 8, 255), (1, 12, 16, 255)]
    for a, b in zip(f32, f64):
        assert np.abs(a - b).max() < 1e-4
        assert np.isfinite(a).all()


def test_upsample_and_decode_semantics(anchors):
    import torch
    t = torch.arange(2 * 3, dtype=torch.float32).view(1, 1, 2, 3)
    up = yolo_ref.upsample_layer(t, (4, 6))[0, 0].numpy()
    np.testing.assert_array_equal(up, np.repeat(np.repeat(t[0, 0].numpy(), 2, 0), 2, 1))
    # decode: zero logits -> centre of the cell, size = anchor, conf = prob = 0.5
    fms = [np.zeros((1, 416 // s, 416 // s, 255), np.float32) for s in (32, 16, 8)]
    boxes, confs, probs = yolo_ref.predict(fms, anchors, [416, 416], 80)
    assert boxes.shape == (1, 10647, 4) and confs.shape == (1, 10647, 1) and probs.shape == (1, 10647, 80)
    assert np.all(confs == 0.5) and np.all(probs == 0.5)
    # first box: scale 13, cell (0,0), anchor 6 = (116, 90): centre (16,16)
    np.testing.assert_allclose(boxes[0, 0], [16 - 58, 16 - 45, 16 + 58, 16 + 45], rtol=1e-6)
    # first box of the 52-grid uses anchor 0 = (10, 13), stride 8
    off = 3 * (13 * 13 + 26 * 26)
    np.testing.assert_allclose(boxes[0, off], [4 - 5, 4 - 6.5, 4 + 5, 4 + 6.5], rtol=1e-6)
    # box index order: (y, x, anchor): second cell in x
    np.testing.assert_allclose(boxes[0, 3], [48 - 58, 16 - 45, 48 + 58, 16 + 45], rtol=1e-6)
