"""GPU parity of the NMS entry points (y3_nms through utils.nms_utils) — bit-exact index selection:
  * py_nms / cpu_nms against the goldens produced by the REFERENCE's own functions;
  * gpu_nms (TF semantics) and cpu_nms against the C oracle on identical inputs at full size
    (10,647 boxes x 80 classes, the thresholds of the reference's call sites), with ties, degenerate
    boxes, empty results, every tier of candidate counts per class, batching;
  * size-independent properties: idempotence, sortedness, max_boxes cap."""
import numpy as np
import pytest
import torch

from conftest import make_boxes

pytestmark = pytest.mark.gpu


def stress_inputs(seed=2, B=10647, C=80, dup=True):
    """SURVEY §8(d) C3 stress: centres U(0,416)^2, w,h U(8,256), scores U*U, plus a duplicated-score subset."""
    rng = np.random.RandomState(seed)
    boxes = make_boxes(rng, B)
    scores = (rng.rand(B, C) * rng.rand(B, C)).astype(np.float32)
    if dup:
        scores[: B // 8] = np.round(scores[: B // 8] * 64) / 64      # many exact ties
        boxes[5:40] = boxes[4]                                       # identical boxes
        boxes[100:120, 2:] = boxes[100:120, :2]                      # zero-area boxes
        boxes[120:140] = boxes[120:140][:, [2, 3, 0, 1]]             # inverted corners
    return boxes, scores


def test_py_nms_reference_goldens(golden):
    from yolov3_tensorflow_amd.utils import nms_utils
    for i in range(int(golden['py_nms_n'])):
        g = {k: golden['py_nms_%d_%s' % (i, k)] for k in ('boxes', 'scores', 'max_boxes', 'iou_thresh', 'keep')}
        keep = nms_utils.py_nms(g['boxes'], g['scores'], int(g['max_boxes']), float(g['iou_thresh']))
        assert keep == g['keep'].tolist(), 'golden case %d' % i


def test_cpu_nms_reference_goldens(golden):
    from yolov3_tensorflow_amd.utils import nms_utils
    for i in range(int(golden['cpu_nms_n'])):
        g = {k: golden['cpu_nms_%d_%s' % (i, k)] for k in
             ('boxes', 'scores', 'num_classes', 'max_boxes', 'score_thresh', 'iou_thresh', 'is_none',
              'out_boxes', 'out_scores', 'out_labels')}
        b, s, l = nms_utils.cpu_nms(g['boxes'][None], g['scores'][None], int(g['num_classes']),
                                    max_boxes=int(g['max_boxes']), score_thresh=float(g['score_thresh']),
                                    iou_thresh=float(g['iou_thresh']))
        if bool(g['is_none']):
            assert b is None and s is None and l is None
        else:
            np.testing.assert_array_equal(b, g['out_boxes'])
            np.testing.assert_array_equal(s, g['out_scores'])
            np.testing.assert_array_equal(l, g['out_labels'])
            assert l.dtype == np.int32


@pytest.mark.parametrize('max_boxes,score_thresh,iou_thresh', [
    (200, 0.3, 0.45),    # test_single_image.py:57
    (400, 0.01, 0.45),   # eval.py:47-54 (~10,000 candidates per class: the sixteen-wave sorted form)
    (150, 0.9, 0.45),    # sparse candidates: the LDS path
    (50, 0.5, 0.5),      # the function defaults
])
@pytest.mark.parametrize('mode', ['tf', 'py'])
def test_full_size_matches_c_oracle(mode, max_boxes, score_thresh, iou_thresh):
    from yolov3_tensorflow_amd.utils import nms_utils
    from yolov3_tensorflow_amd import _lib
    from oracle import nms_ref
    boxes, scores = stress_inputs()
    ob, osc, ol, oi = nms_ref.c_per_class(mode, boxes, scores, 80, max_boxes, score_thresh, iou_thresh)
    m = _lib.Y3_NMS_TF if mode == 'tf' else _lib.Y3_NMS_PY
    b = torch.from_numpy(boxes).cuda()[None]
    s = torch.from_numpy(scores).cuda()[None]
    gb, gs, gl, gi, cnt = nms_utils._run_nms(m, b, s, 80, max_boxes, score_thresh, iou_thresh)
    k = int(cnt[0])
    assert k == len(ob), 'selected %d, oracle %d' % (k, len(ob))
    np.testing.assert_array_equal(gi[0, :k].cpu().numpy(), oi)      # bit-exact index selection
    np.testing.assert_array_equal(gl[0, :k].cpu().numpy(), ol)
    np.testing.assert_array_equal(gb[0, :k].cpu().numpy(), ob)
    np.testing.assert_array_equal(gs[0, :k].cpu().numpy(), osc)


@pytest.mark.parametrize('mode', ['tf', 'py'])
def test_every_tier_of_candidate_counts(mode):
    """Three kernels by the number of candidates K of an (image, class): the sorted form on ONE wave (K <= 512), on sixteen
    waves (K <= 16,384: keys in the LDS), and the arg-max form on global memory beyond (a 608x608 image has 22,743 boxes).
    One call with all of them - class 0 keeps every one of 20,000 boxes, class 1 about 7,000, class 2 about 600 (just past
    the one-wave form), class 3 about 200 - bit-exact against the C oracle, with max_boxes below and above what the classes
    can fill (early exit and exhaustion)."""
    from yolov3_tensorflow_amd.utils import nms_utils
    from yolov3_tensorflow_amd import _lib
    from oracle import nms_ref
    boxes, scores = stress_inputs(seed=7, B=20000, C=4)
    scores[:, 0] = 0.5 + 0.5 * scores[:, 0]            # all above the threshold
    scores[:, 1] = np.where(scores[:, 1] > 0.3, scores[:, 1], 0.1)
    scores[:, 2] = np.where(scores[:, 2] > 0.75, scores[:, 2], 0.1)
    scores[:, 3] = np.where(scores[:, 3] > 0.85, scores[:, 3], 0.1)
    counts = [int((scores[:, c] >= 0.2).sum()) for c in range(4)]
    assert counts[0] > 16384 >= counts[1] > 512 and 16384 >= counts[2] > 512 >= counts[3] > 0, counts
    m = _lib.Y3_NMS_TF if mode == 'tf' else _lib.Y3_NMS_PY
    for max_boxes in (40, 300, 1500, 2500):      # (2,500 selected boxes do not fit the LDS beside the keys: arg-max form for all)
        ob, osc, ol, oi = nms_ref.c_per_class(mode, boxes, scores, 4, max_boxes, 0.2, 0.45)
        gb, gs, gl, gi, cnt = nms_utils._run_nms(m, torch.from_numpy(boxes).cuda()[None], torch.from_numpy(scores).cuda()[None],
                                                 4, max_boxes, 0.2, 0.45)
        k = int(cnt[0])
        assert k == len(ob), 'max_boxes %d: selected %d, oracle %d' % (max_boxes, k, len(ob))
        np.testing.assert_array_equal(gi[0, :k].cpu().numpy(), oi)
        np.testing.assert_array_equal(gl[0, :k].cpu().numpy(), ol)
        np.testing.assert_array_equal(gs[0, :k].cpu().numpy(), osc)


def test_gpu_nms_api_and_empty_result():
    from yolov3_tensorflow_amd.utils import nms_utils
    boxes, scores = stress_inputs(seed=3, B=2000, C=7, dup=False)
    b, s, l = nms_utils.gpu_nms(boxes[None], scores[None], 7, max_boxes=20, score_thresh=0.2, nms_thresh=0.45)
    assert b.dim() == 2 and b.shape[1] == 4 and s.shape == l.shape == (b.shape[0],)
    assert l.dtype == torch.int32 and b.shape[0] > 0
    # labels ascending, scores non-increasing inside a class, at most max_boxes per class
    ln, sn = l.cpu().numpy(), s.cpu().numpy()
    assert (np.diff(ln) >= 0).all()
    for c in range(7):
        sc = sn[ln == c]
        assert len(sc) <= 20 and (np.diff(sc) <= 0).all()
    # nothing above the threshold: gpu_nms -> empty tensors, cpu_nms -> (None, None, None)
    b, s, l = nms_utils.gpu_nms(boxes[None], scores[None], 7, score_thresh=2.0)
    assert tuple(b.shape) == (0, 4) and tuple(s.shape) == (0,) and tuple(l.shape) == (0,)
    assert nms_utils.cpu_nms(boxes[None], scores[None], 7, score_thresh=2.0) == (None, None, None)
    assert nms_utils.py_nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32)) == []


def test_idempotence_at_full_size():
    """TF-mode NMS of its own output selects everything again, in the same order."""
    from yolov3_tensorflow_amd.utils import nms_utils
    boxes, scores = stress_inputs(dup=False)
    res = nms_utils.gpu_nms_batched(boxes[None], scores[None], 80, 200, 0.3, 0.45, return_index=True)[0]
    b1, s1, l1, i1 = res
    K = b1.shape[0]
    assert K > 100
    # rebuild a [K, 80] score matrix holding each survivor's score in its own class
    sc = torch.zeros((K, 80), device='cuda')
    sc[torch.arange(K, device='cuda'), l1.long()] = s1
    b2, s2, l2 = nms_utils.gpu_nms(b1[None], sc[None], 80, 200, 0.3, 0.45)
    assert torch.equal(b2, b1) and torch.equal(s2, s1) and torch.equal(l2, l1)


def test_batched_equals_per_image():
    from yolov3_tensorflow_amd.utils import nms_utils
    bs, ss = zip(*[stress_inputs(seed=10 + i, B=3000, C=12) for i in range(3)])
    boxes, scores = np.stack(bs), np.stack(ss)
    batched = nms_utils.gpu_nms_batched(boxes, scores, 12, 30, 0.25, 0.45, return_index=True)
    for i in range(3):
        b, s, l = nms_utils.gpu_nms(boxes[i:i + 1], scores[i:i + 1], 12, 30, 0.25, 0.45)
        assert torch.equal(batched[i][0], b) and torch.equal(batched[i][1], s) and torch.equal(batched[i][2], l)


def test_tie_break_and_threshold_edges():
    from yolov3_tensorflow_amd.utils import nms_utils
    # equal scores: lower index wins; IoU exactly at the threshold is kept by TF (>) and by py (<=)
    boxes = np.array([[0, 0, 2, 2], [0, 1, 2, 3], [10, 10, 12, 12], [10, 10, 12, 12]], np.float32)
    scores = np.array([[0.5], [0.5], [0.7], [0.7]], np.float32)
    thr = float(np.float32(2.0) / np.float32(6.0))
    out = nms_utils.gpu_nms_batched(boxes[None], scores[None], 1, 10, 0.1, thr, return_index=True)[0]
    assert out[3].cpu().tolist() == [2, 0, 1]
    # score exactly at the threshold passes (>=)
    b, s, l = nms_utils.gpu_nms(boxes[None], scores[None], 1, 10, 0.7, 0.5)
    assert s.cpu().tolist() == [pytest.approx(0.7)]


def test_nms_argument_validation():
    from yolov3_tensorflow_amd.utils import nms_utils
    boxes, scores = stress_inputs(B=100, C=3)
    with pytest.raises(ValueError):
        nms_utils.gpu_nms(boxes[None], scores[None], 3, max_boxes=0)
