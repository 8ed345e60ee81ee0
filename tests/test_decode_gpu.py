"""GPU parity of yolov3.predict / reorg_layer (y3_decode, y3_reorg_boxes) against the numpy oracle.
Tolerance (stated): sigmoid outputs |d| <= 2e-7 abs + 2e-6 rel; box corners 4e-6 relative to the box's own
scale max(|corner|) (x_min = cx - w/2 cancels, so the error scales with w, not with x_min; expf vs numpy
exp differ by <= 2 ulp); scores == confs*probs bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def box_close(got, want, rtol=4e-6, atol=1e-5):
    scale = np.abs(want).max(axis=-1, keepdims=True)
    return bool((np.abs(got - want) <= atol + rtol * scale).all())


def _model(anchors, class_num=80, hw=(416, 416)):
    import yolov3_tensorflow_amd as y3
    m = y3.yolov3(class_num, anchors)
    m.img_size = list(hw)
    return m


def _fms(rng, n, h, w, C, std=2.0):
    ch = 3 * (5 + C)
    return [(rng.standard_normal((n, h // s, w // s, ch)) * std).astype(np.float32) for s in (32, 16, 8)]


@pytest.mark.parametrize('n,h,w,C', [(2, 416, 416, 80), (1, 608, 608, 80), (3, 320, 480, 20), (1, 64, 32, 1),
                                     # large heads: 248 is the last class count on the LDS-staged kernel, 252 and 600
                                     # need more than 64 KB of LDS per workgroup and take the 4-byte kernel
                                     (1, 96, 64, 248), (1, 96, 64, 252), (2, 64, 64, 600)])
def test_predict_matches_oracle(anchors, n, h, w, C):
    from oracle import yolo_ref
    rng = np.random.RandomState(h + C)
    fms = _fms(rng, n, h, w, C)
    m = _model(anchors, C, (h, w))
    boxes, confs, probs, scores = m.predict(fms, with_scores=True)
    rb, rc, rp = yolo_ref.predict(fms, anchors, [h, w], C)
    B = 3 * sum((h // s) * (w // s) for s in (32, 16, 8))
    assert tuple(boxes.shape) == (n, B, 4) and tuple(confs.shape) == (n, B, 1) and tuple(probs.shape) == (n, B, C)
    b = boxes.cpu().numpy()
    assert box_close(b, rb), np.abs(b - rb).max()
    np.testing.assert_allclose(confs.cpu().numpy(), rc, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(probs.cpu().numpy(), rp, rtol=2e-6, atol=2e-7)
    assert torch.equal(scores, confs * probs)
    # the 3-tuple form of the reference API
    b3 = m.predict(fms)
    assert len(b3) == 3 and torch.equal(b3[0], boxes)


def test_predict_full_batch_known_answers(anchors):
    """bs=32 @416 with zero logits: every box is its anchor centred on its cell, conf = prob = 0.5."""
    m = _model(anchors)
    fms = [torch.zeros((32, 416 // s, 416 // s, 255), device='cuda') for s in (32, 16, 8)]
    boxes, confs, probs = m.predict(fms)
    assert torch.all(confs == 0.5) and torch.all(probs == 0.5)
    b = boxes.cpu().numpy()
    wh = b[..., 2:] - b[..., :2]
    ctr = (b[..., 2:] + b[..., :2]) / 2
    off = 0
    for s, a0 in ((32, 6), (16, 3), (8, 0)):
        g = 416 // s
        blk_wh = wh[:, off:off + g * g * 3].reshape(32, g, g, 3, 2)
        blk_c = ctr[:, off:off + g * g * 3].reshape(32, g, g, 3, 2)
        np.testing.assert_allclose(blk_wh, np.broadcast_to(anchors[a0:a0 + 3], blk_wh.shape), rtol=1e-5)
        gx = (np.arange(g) + 0.5) * s
        np.testing.assert_allclose(blk_c[..., 0], np.broadcast_to(gx[None, None, :, None], blk_c[..., 0].shape), rtol=1e-6)
        np.testing.assert_allclose(blk_c[..., 1], np.broadcast_to(gx[None, :, None, None], blk_c[..., 1].shape), rtol=1e-6)
        off += g * g * 3
    assert off == 10647


def test_reorg_layer_matches_oracle(anchors):
    from oracle import yolo_ref
    rng = np.random.RandomState(1)
    fm = (rng.standard_normal((2, 26, 26, 255)) * 1.5).astype(np.float32)
    m = _model(anchors)
    xy, boxes, conf_logits, prob_logits = m.reorg_layer(fm, anchors[3:6])
    rxy, rboxes, rconf, rprob = yolo_ref.reorg_layer(fm, anchors[3:6], [416, 416], 80)
    np.testing.assert_array_equal(xy.cpu().numpy(), rxy)
    rb = rboxes
    b = boxes.cpu().numpy()
    assert b.shape == (2, 26, 26, 3, 4)
    assert box_close(b, rb)
    np.testing.assert_array_equal(conf_logits.cpu().numpy(), rconf)
    np.testing.assert_array_equal(prob_logits.cpu().numpy(), rprob)


def test_predict_rejects_mismatched_maps(anchors):
    m = _model(anchors)
    fms = [torch.zeros((1, 13, 13, 255), device='cuda'), torch.zeros((1, 26, 26, 255), device='cuda'),
           torch.zeros((1, 26, 26, 255), device='cuda')]
    with pytest.raises(ValueError):
        m.predict(fms)
    with pytest.raises(ValueError):
        m.predict([torch.zeros((1, 13, 13, 18), device='cuda')] * 3)
