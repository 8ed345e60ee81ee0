"""Parity at the BENCH configuration (BASELINE configs[1]/[2]: bs=32, 416x416, the default 'f32_wino' mode): the
stream-K split points — hence the fp32 summation order and the in-kernel hand-off paths the bench actually runs —
depend on the batch size, so the reduced-batch tests do not cover them.

  forward -> predict (+ conf*prob) -> gpu_nms for all 32 images (ref: model.py:30-190, utils/nms_utils.py:8-48)
    * all 32 images vs the CPU fp32 oracle, the first 4 also vs the fp64 oracle, at the tolerances of
      tests/test_forward_gpu.py (feature maps |d| <= 2e-4 + 1e-4*|ref|) and tests/test_pipeline_gpu.py
      (confs/probs 1e-3; box corners 1e-3 px + 1e-3 * the box's own scale);
    * NMS selection bit-exact against the C oracle on the GPU's own decoded boxes/scores, every image;
    * the direct-kernel mode ('f32') on the same batch at the same tolerance.

A second, independent convolution engine — torch.nn.functional.conv2d on ROCm, i.e. MIOpen (SURVEY §8c allows a
secondary cross-check on the GPU box; it is TEST-ONLY and never the product) — is compared with the HIP conv on every
distinct conv shape of the network, so that two unrelated implementations (oneDNN on the CPU in the oracle, MIOpen on
the GPU here) agree with the hand-written kernels.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import blob_images

pytestmark = pytest.mark.gpu

BATCH, SIZE = 32, 416


def _fm_check(got, want, what, atol=2e-4, rtol=1e-4):
    err = np.abs(got - want)
    assert got.shape == want.shape
    assert np.isfinite(got).all(), what
    assert (err <= atol + rtol * np.abs(want)).all(), '%s: max err %.3e' % (what, err.max())
    return float(err.max())


@pytest.fixture(scope='module')
def bench_batch(gpu_model):
    """The bs=32 batch, the oracle's feature maps for it (fp32: all images; fp64: the first 4)."""
    from oracle import yolo_ref
    model, params = gpu_model
    x = blob_images(77, BATCH, SIZE)
    torch.set_num_threads(min(torch.get_num_threads(), 64))
    ref32 = [np.concatenate(parts) for parts in zip(*[yolo_ref.forward(params, x[i:i + 8]) for i in range(0, BATCH, 8)])]
    ref64 = yolo_ref.forward(params, x[:4], dtype=torch.float64)
    return x, ref32, ref64


@pytest.mark.parametrize('dtype', ['f32_wino', 'f32'])
def test_bs32_416_forward_predict_nms_matches_oracle(gpu_model, anchors, bench_batch, dtype):
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.utils import nms_utils
    from oracle import yolo_ref, nms_ref
    model, params = gpu_model
    x, ref32, ref64 = bench_batch
    model.compute_dtype = dtype
    try:
        with y3.variable_scope('yolov3'):
            fms = model.forward(x, False)
            fms_again = model.forward(x, False)
    finally:
        model.compute_dtype = 'f32'
    for a, b in zip(fms, fms_again):
        assert torch.equal(a, b), 'bs=32 forward is not run-to-run bit-exact'
    worst32 = worst64 = 0.0
    for i, (g, r32, r64) in enumerate(zip(fms, ref32, ref64)):
        g = g.cpu().numpy()
        worst32 = max(worst32, _fm_check(g, r32, '%s feature_map_%d (32 images) vs fp32 oracle' % (dtype, i + 1)))
        worst64 = max(worst64, _fm_check(g[:4], r64, '%s feature_map_%d (4 images) vs fp64 oracle' % (dtype, i + 1)))
    print('%s bs=32: feature maps max |d| vs fp32 oracle %.3e, vs fp64 oracle (4 images) %.3e' % (dtype, worst32, worst64))

    boxes, confs, probs, scores = model.predict(fms, with_scores=True)
    rb, rc, rp = yolo_ref.predict(ref32, anchors, [SIZE, SIZE], 80)
    gb, gc, gp, gs = (t.cpu().numpy() for t in (boxes, confs, probs, scores))
    assert np.abs(gc - rc).max() <= 1e-3 and np.abs(gp - rp).max() <= 1e-3
    scale = np.abs(rb).max(axis=-1, keepdims=True)
    berr = np.abs(gb - rb)
    assert (berr <= 1e-3 + 1e-3 * scale).all(), 'boxes: max err %.3e' % berr.max()
    print('%s bs=32: box max |d| %.3e px (rel. to box scale %.3e), confs %.3e, probs %.3e' %
          (dtype, berr.max(), (berr / np.maximum(scale, 1.0)).max(), np.abs(gc - rc).max(), np.abs(gp - rp).max()))

    # NMS (eval-like regime: ~6000 candidates per image) for the whole batch in one launch set
    thr = float(np.quantile(gs[0], 1 - 6000.0 / gs[0].size))
    out = nms_utils.gpu_nms_batched(boxes, scores, 80, 100, thr, 0.45, return_index=True)
    rs = rc * rp
    agree_min = 1.0
    for i in range(BATCH):
        b, s, l, idx = out[i]
        ob, osc, ol, oi = nms_ref.c_per_class('tf', gb[i], gs[i], 80, 100, thr, 0.45)
        np.testing.assert_array_equal(idx.cpu().numpy(), oi)          # identical inputs: bit-exact selection
        np.testing.assert_array_equal(l.cpu().numpy(), ol)
        np.testing.assert_array_equal(b.cpu().numpy(), ob)
        np.testing.assert_array_equal(s.cpu().numpy(), osc)
        if i < 8:                                                      # end to end vs the oracle's own features
            eb, es, el, ei = nms_ref.c_per_class('tf', rb[i], rs[i], 80, 100, thr, 0.45)
            got, want = set(zip(ol.tolist(), oi.tolist())), set(zip(el.tolist(), ei.tolist()))
            agree_min = min(agree_min, len(got & want) / float(max(len(want), 1)))
    print('%s bs=32: end-to-end NMS index agreement (worst of 8 images) %.4f' % (dtype, agree_min))
    assert agree_min >= 0.99


# (k, stride, cin, cout): every distinct conv shape of the graph + the 20-class detection conv
SHAPES = [
    (3, 1, 3, 32), (3, 2, 32, 64), (1, 1, 64, 32), (3, 1, 32, 64), (3, 2, 64, 128), (1, 1, 128, 64), (3, 1, 64, 128),
    (3, 2, 128, 256), (1, 1, 256, 128), (3, 1, 128, 256), (3, 2, 256, 512), (1, 1, 512, 256), (3, 1, 256, 512),
    (3, 2, 512, 1024), (1, 1, 1024, 512), (3, 1, 512, 1024), (1, 1, 1024, 255), (1, 1, 512, 255), (1, 1, 256, 255),
    (1, 1, 768, 256), (1, 1, 384, 128), (1, 1, 512, 256), (1, 1, 256, 128), (3, 1, 256, 512), (3, 1, 128, 256),
    (1, 1, 1024, 75),
]


@pytest.mark.parametrize('k,stride,cin,cout', SHAPES)
def test_hip_conv_agrees_with_miopen(k, stride, cin, cout):
    """Secondary cross-check (test-only): torch's ROCm conv2d (MIOpen, fp32) vs y3_conv2d_fwd, and for the eligible
    shapes vs y3_conv2d_fwd_wino, on the same device tensors.  Tolerance 1e-4*(1+|ref|) — the bound of
    tests/test_conv_gpu.py — here between two fp32 implementations with different summation orders."""
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    rng = np.random.RandomState(cin * 7 + cout + k + stride)
    n, h, w = 4, 26, 38
    x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).to(dev)
    wt = torch.from_numpy((rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)).to(dev)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
    shift = torch.from_numpy(rng.normal(0, 0.2, cout).astype(np.float32)).to(dev)
    xn = x.permute(0, 3, 1, 2).contiguous()
    wo = wt.permute(3, 2, 0, 1).contiguous()
    prev = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    try:
        if stride > 1:
            y = F.conv2d(F.pad(xn, (1, 1, 1, 1)), wo, stride=stride)
        else:
            y = F.conv2d(xn, wo, padding=k // 2)
    finally:
        torch.backends.cudnn.allow_tf32 = prev
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    want = torch.where(y > 0, y, 0.1 * y).permute(0, 2, 3, 1).contiguous().cpu().numpy()

    if cin == 3:
        wp = wt
    else:
        wp = torch.empty(k * k * cout * cin, device=dev)
        _lib.check(_lib.lib().y3_pack_conv_weights(fw.context(), fw.ptr(wt), k, cin, cout, fw.ptr(wp)))
    got = engine.conv2d_fwd(x, wp, scale, shift, k, stride, cout, True).cpu().numpy()
    err = np.abs(got - want)
    assert (err <= 1e-4 * (1 + np.abs(want))).all(), 'direct kernel vs MIOpen: max err %.3e' % err.max()
    if engine.wino_eligible(k, stride, cin, cout):
        got = engine.conv2d_fwd_wino(x, engine.pack_wino(wt), scale, shift, cout, True).cpu().numpy()
        err = np.abs(got - want)
        assert (err <= 1e-4 * (1 + np.abs(want))).all(), 'Winograd kernel vs MIOpen: max err %.3e' % err.max()
