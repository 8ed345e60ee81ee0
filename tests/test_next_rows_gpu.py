"""GPU tests of the SURVEY §8(f) 'next' rows built so far: device-side target assignment (`process_box`,
bit-exact against the reference's own function through the golden vectors) and the batched detect path."""
import numpy as np
import pytest
import torch

from conftest import COCO_ANCHORS, blob_images

pytestmark = pytest.mark.gpu


def test_process_box_matches_reference_goldens_bit_exact(golden):
    from yolov3_tensorflow_amd.utils import data_utils
    for i in range(int(golden['process_box_n'])):
        boxes, labels = golden['process_box_%d_boxes' % i], golden['process_box_%d_labels' % i]
        ys = data_utils.process_box(boxes, labels, [416, 416], 80, COCO_ANCHORS)
        for y, k in zip(ys, ('y13', 'y26', 'y52')):
            np.testing.assert_array_equal(y, golden['process_box_%d_%s' % (i, k)])


def test_process_box_batch_ragged_and_overwrite_semantics():
    from yolov3_tensorflow_amd.utils import data_utils
    from oracle import train_ref
    rng = np.random.RandomState(1)
    n, kmax = 5, 12
    boxes = np.zeros((n, kmax, 5), np.float32)
    labels = np.zeros((n, kmax), np.int64)
    counts = [0, 1, 12, 7, 3]
    for i, K in enumerate(counts):
        wh = rng.uniform(10, 300, (K, 2)); c = rng.uniform(150, 266, (K, 2))
        boxes[i, :K] = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(0.2, 1.0, (K, 1))], 1)
        labels[i, :K] = rng.randint(0, 80, K)
    boxes[2, 5] = boxes[2, 4]; labels[2, 5] = (labels[2, 4] + 1) % 80      # same cell/anchor: overwrite, multi-hot
    ys = data_utils.process_box_batch(boxes, labels, counts, [416, 416], 80, COCO_ANCHORS)
    for i, K in enumerate(counts):
        if K == 0:
            for y in ys:
                got = y[i].cpu().numpy()
                assert (got[..., :-1] == 0).all() and (got[..., -1] == 1).all()
            continue
        ref = train_ref.process_box(boxes[i, :K], labels[i, :K], [416, 416], 80, COCO_ANCHORS)
        for y, r in zip(ys, ref):
            np.testing.assert_array_equal(y[i].cpu().numpy(), r)


def test_detect_equals_forward_predict_nms(gpu_model):
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.utils import nms_utils
    model, _ = gpu_model
    x = blob_images(17, 3, 256)
    with y3.variable_scope('yolov3'):
        fms = model.forward(x, False)
        boxes, confs, probs = model.predict(fms)
        thr = float(torch.quantile((confs * probs).flatten()[::7], 0.999))
        dets = model.detect(x, max_boxes=50, score_thresh=thr, nms_thresh=0.45)
    assert len(dets) == 3
    for i, (b, s, l) in enumerate(dets):
        rb, rs, rl = nms_utils.gpu_nms(boxes[i:i + 1], (confs * probs)[i:i + 1], 80, 50, thr, 0.45)
        assert torch.equal(b, rb) and torch.equal(s, rs) and torch.equal(l, rl)
        assert b.shape[0] > 0
