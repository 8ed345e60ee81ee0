"""GPU parity of the training path against the torch-autograd oracle (oracle/train_ref.py, fp64):
conv weight/data gradients, batch-norm train forward/backward, the YOLO loss and its gradient, and one
whole train step (loss 5-tuple, clipped gradients, updated variables, BN moving statistics) for each of
the four optimizers.  Tolerances (stated): per-op 2e-4 relative to the tensor's max magnitude; whole-step
gradients 2e-4 relative to the tensor's max |grad| for EVERY tensor (256 px, bs=4, fp64 oracle evaluated on the
LeakyReLU branches the GPU took — see test_one_train_step_matches_oracle); loss values 1e-4 relative."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import COCO_ANCHORS, blob_images

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-12))


def _ctx():
    from yolov3_tensorflow_amd import framework as fw, _lib
    return fw, _lib, _lib.lib(), fw.context()


@pytest.mark.parametrize('n,h,w,k,stride,cin,cout', [
    (2, 20, 28, 3, 1, 64, 128), (2, 20, 28, 1, 1, 128, 64), (3, 16, 24, 3, 2, 32, 64),
    (2, 13, 13, 1, 1, 256, 255), (2, 26, 26, 3, 1, 32, 64), (1, 40, 40, 3, 1, 3, 32), (2, 12, 12, 3, 2, 128, 256),
    # the stem's weight gradient walks 128-pixel pieces of image rows: a last piece of 32 pixels (416), of 2 (130), none (128)
    (2, 24, 416, 3, 1, 3, 32), (1, 9, 130, 3, 1, 3, 32), (3, 5, 128, 3, 1, 3, 32),
])
def test_conv_wgrad_and_dgrad_match_autograd(n, h, w, k, stride, cin, cout):
    fw, _lib, L, ctx = _ctx()
    dev = fw.default_device()
    rng = np.random.RandomState(cin + cout + k)
    x = torch.tensor(rng.standard_normal((n, h, w, cin)), dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(rng.standard_normal((k, k, cin, cout)) * 0.1, dtype=torch.float64, requires_grad=True)
    xp = x.permute(0, 3, 1, 2)
    if stride > 1:
        xp = F.pad(xp, (1, 1, 1, 1))
        z = F.conv2d(xp, wt.permute(3, 2, 0, 1), stride=stride)
    else:
        z = F.conv2d(xp, wt.permute(3, 2, 0, 1), padding=k // 2)
    ho, wo = z.shape[2], z.shape[3]
    dz = rng.standard_normal((n, ho, wo, cout))
    z.backward(torch.tensor(dz).permute(0, 3, 1, 2))
    stride_c = ((cout + 31) // 32) * 32
    dzp = np.zeros((n, ho, wo, stride_c), np.float32)
    dzp[..., :cout] = dz
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, k, stride, 0)
    xg = torch.tensor(x.detach().numpy(), dtype=torch.float32, device=dev)
    dzg = torch.from_numpy(dzp).to(dev)
    wg = torch.tensor(wt.detach().numpy(), dtype=torch.float32, device=dev)
    # weight gradient
    dw = torch.empty((k, k, cin, cout), device=dev)
    sc = torch.empty(L.y3_conv_wgrad_scratch_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    _lib.check(L.y3_conv_wgrad(ctx, ctypes.byref(d), fw.ptr(xg), fw.ptr(dzg), stride_c, fw.ptr(dw), fw.ptr(sc),
                               ctypes.c_size_t(sc.numel())))
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 2e-4
    if cin == 3:
        return
    # data gradient (overwrite, then accumulate on top)
    w_d = torch.zeros((k * k * cin, stride_c), device=dev)
    w_d[:, :cout] = wg.reshape(k * k * cin, cout)
    ones, zeros = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    dx = torch.empty((n, h, w, cin), device=dev)
    ws = torch.empty(512 * 2 * 128 * 128 * 4, dtype=torch.uint8, device=dev)
    for acc, mult in ((0, 1.0), (1, 2.0)):
        _lib.check(L.y3_conv2d_dgrad(ctx, ctypes.byref(d), fw.ptr(dzg), stride_c, fw.ptr(w_d), fw.ptr(ones),
                                     fw.ptr(zeros), acc, fw.ptr(dx), fw.ptr(ws), ctypes.c_size_t(ws.numel())))
        assert rel_err(dx.cpu().numpy(), mult * x.grad.numpy()) < 2e-4, 'accumulate=%d' % acc
    # the same data gradient on the bf16 matrix pipe (stride-1 layers), same tolerance
    if stride == 1 and cin % 4 == 0:
        for planes, tol in ((3, 2e-4), (2, 5e-3)):
            wsd = torch.empty(planes * k * k * cin * stride_c, dtype=torch.bfloat16, device=dev)
            _lib.check(L.y3_pack_conv_weights_split_dgrad(ctx, fw.ptr(w_d), k, cin, stride_c, planes, fw.ptr(wsd)))
            for acc, mult in ((0, 1.0), (1, 2.0)):
                _lib.check(L.y3_conv2d_dgrad_split(ctx, ctypes.byref(d), planes, fw.ptr(dzg), stride_c, fw.ptr(wsd),
                                                   fw.ptr(ones), fw.ptr(zeros), acc, fw.ptr(dx), fw.ptr(ws),
                                                   ctypes.c_size_t(ws.numel())))
                assert rel_err(dx.cpu().numpy(), mult * x.grad.numpy()) < tol, 'split planes=%d accumulate=%d' % (planes, acc)
    else:
        with pytest.raises(ValueError):
            _lib.check(L.y3_conv2d_dgrad_split(ctx, ctypes.byref(d), 3, fw.ptr(dzg), stride_c, fw.ptr(w_d),
                                               fw.ptr(ones), fw.ptr(zeros), 0, fw.ptr(dx), fw.ptr(ws),
                                               ctypes.c_size_t(ws.numel())))
    # the same data gradient in Winograd form (stride-1 3x3 convs): the gradient conv is itself a 3x3 SAME conv
    if k == 3 and stride == 1 and cin % 32 == 0:
        wwd = torch.empty(16 * cin * stride_c, device=dev)
        _lib.check(L.y3_pack_conv_weights_wino_dgrad(ctx, fw.ptr(w_d), cin, stride_c, fw.ptr(wwd)))
        for use_ws in (True, False):
            for acc, mult in ((0, 1.0), (1, 2.0)):
                _lib.check(L.y3_conv2d_dgrad_wino(ctx, ctypes.byref(d), fw.ptr(dzg), stride_c, fw.ptr(wwd), fw.ptr(ones),
                                                  fw.ptr(zeros), acc, fw.ptr(dx), fw.ptr(ws) if use_ws else None,
                                                  ctypes.c_size_t(ws.numel() if use_ws else 0)))
                assert rel_err(dx.cpu().numpy(), mult * x.grad.numpy()) < 2e-4, 'wino accumulate=%d ws=%s' % (acc, use_ws)
    elif stride == 2:
        with pytest.raises(ValueError):
            _lib.check(L.y3_conv2d_dgrad_wino(ctx, ctypes.byref(d), fw.ptr(dzg), stride_c, fw.ptr(w_d), fw.ptr(ones),
                                              fw.ptr(zeros), 0, fw.ptr(dx), None, ctypes.c_size_t(0)))


@pytest.mark.parametrize('n,h,w,cin,cout', [
    (2, 20, 28, 64, 128),      # even map, several K-steps, two column blocks
    (1, 13, 13, 128, 64),      # odd map: the last tile row / column has outputs that do not exist
    (3, 7, 5, 64, 64),         # tiny odd map, T = 36 tiles: a partial last K-step
    (2, 26, 26, 256, 128),     # 8 workgroup tiles x 32 splits
    (2, 16, 4, 64, 64),        # two tile columns: a K-step advances four tile rows
    (8, 52, 52, 128, 256),     # a real layer shape (52-grid residual stage) at bs=8: 676 K-steps over 32 splits
])
def test_winograd_weight_gradient_matches_autograd(n, h, w, cin, cout):
    """y3_conv_wgrad_wino: dw = G^T [ sum over tiles (B^T d B) .* (A dY A^T) ] G against fp64 autograd (ref: train.py:112,
    the kernel gradients of utils/layer_utils.py:9-22) at the direct weight-gradient kernel's tolerance (2e-4 of the
    tensor's max), run-to-run bit-exact, and interchangeable with y3_conv_wgrad."""
    fw, _lib, L, ctx = _ctx()
    dev = fw.default_device()
    rng = np.random.RandomState(cin + cout + h)
    x = torch.tensor(rng.standard_normal((n, h, w, cin)), dtype=torch.float64)
    wt = torch.zeros((3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    z = F.conv2d(x.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), padding=1)
    dz = rng.standard_normal((n, h, w, cout))
    z.backward(torch.tensor(dz).permute(0, 3, 1, 2))
    want = wt.grad.numpy()
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, 3, 1, 0)
    assert L.y3_conv_wgrad_wino_eligible(ctypes.byref(d)) == 1
    xg = torch.tensor(x.numpy(), dtype=torch.float32, device=dev)
    dzg = torch.tensor(dz, dtype=torch.float32, device=dev)
    sc = torch.empty(L.y3_conv_wgrad_wino_scratch_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        dw = torch.full((3, 3, cin, cout), float('nan'), device=dev)
        _lib.check(L.y3_conv_wgrad_wino(ctx, ctypes.byref(d), fw.ptr(xg), fw.ptr(dzg), cout, fw.ptr(dw), fw.ptr(sc),
                                        ctypes.c_size_t(sc.numel())))
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]), 'not run-to-run bit-exact'
    e = rel_err(outs[0].cpu().numpy(), want)
    assert e < 2e-4, 'Winograd wgrad rel err %.3e' % e
    # the direct kernel on the same operands: the two must agree to the same tolerance
    dw2 = torch.empty((3, 3, cin, cout), device=dev)
    sc2 = torch.empty(L.y3_conv_wgrad_scratch_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    _lib.check(L.y3_conv_wgrad(ctx, ctypes.byref(d), fw.ptr(xg), fw.ptr(dzg), cout, fw.ptr(dw2), fw.ptr(sc2),
                               ctypes.c_size_t(sc2.numel())))
    assert rel_err(outs[0].cpu().numpy(), dw2.cpu().numpy()) < 2e-4
    # a padded dz row stride (the detection convs pad 255 -> 256; here 3x3 layers with extra columns)
    dzp = torch.zeros((n, h, w, cout + 32), dtype=torch.float32, device=dev)
    dzp[..., :cout] = dzg
    dw3 = torch.empty((3, 3, cin, cout), device=dev)
    _lib.check(L.y3_conv_wgrad_wino(ctx, ctypes.byref(d), fw.ptr(xg), fw.ptr(dzp), cout + 32, fw.ptr(dw3), fw.ptr(sc),
                                    ctypes.c_size_t(sc.numel())))
    assert torch.equal(dw3, outs[0])
    # shapes the kernel does not take are refused, not mangled
    for bad in (_lib.ConvDesc(n, h, w, 32, 0, cout, 3, 1, 0), _lib.ConvDesc(n, h, w, cin, 0, cout, 3, 2, 0),
                _lib.ConvDesc(n, h, w, cin, 0, cout, 1, 1, 0), _lib.ConvDesc(n, 2, 2, cin, 0, cout, 3, 1, 0)):
        assert L.y3_conv_wgrad_wino_eligible(ctypes.byref(bad)) == 0
        with pytest.raises(ValueError):
            _lib.check(L.y3_conv_wgrad_wino(ctx, ctypes.byref(bad), fw.ptr(xg), fw.ptr(dzg), cout, fw.ptr(dw3), fw.ptr(sc),
                                            ctypes.c_size_t(sc.numel())))


@pytest.mark.parametrize('n,h,w,cin,cout', [(2, 13, 13, 64, 128), (4, 52, 52, 128, 256), (1, 20, 28, 256, 96)])
def test_winograd_f4x4_data_gradient_matches_autograd(n, h, w, cin, cout):
    """y3_conv2d_dgrad_wino44: the data gradient of a stride-1 3x3 SAME conv (ref: TF autodiff of slim.conv2d, train.py:112)
    as the F(4x4,3x3) kernel on dz with the flipped, channel-swapped kernel, against fp64 autograd (2e-4 of the gradient's
    max magnitude, the tolerance of the other data-gradient kernels), plain and accumulating into an existing gradient."""
    from yolov3_tensorflow_amd import engine, framework as fw
    dev = fw.default_device()
    rng = np.random.RandomState(n + h + cin + cout)
    x = rng.standard_normal((n, h, w, cin))
    wt = rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))
    dz = rng.standard_normal((n, h, w, cout))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(xt.permute(0, 3, 1, 2), torch.tensor(wt).permute(3, 2, 0, 1), padding=1)
    y.backward(torch.tensor(dz).permute(0, 3, 1, 2))
    want = xt.grad.numpy()
    got = engine.conv2d_dgrad_wino44(torch.tensor(dz, dtype=torch.float32, device=dev),
                                     torch.tensor(wt, dtype=torch.float32, device=dev), cin)
    scale = np.abs(want).max()
    assert np.abs(got.cpu().numpy() - want).max() <= 2e-4 * scale
    base = torch.tensor(rng.standard_normal((n, h, w, cin)), dtype=torch.float32, device=dev)
    acc = engine.conv2d_dgrad_wino44(torch.tensor(dz, dtype=torch.float32, device=dev),
                                     torch.tensor(wt, dtype=torch.float32, device=dev), cin, accumulate_into=base.clone())
    assert np.abs(acc.cpu().numpy() - (want + base.cpu().numpy())).max() <= 2e-4 * scale + 1e-6
    one = engine.conv2d_dgrad_wino44(torch.tensor(dz, dtype=torch.float32, device=dev),
                                     torch.tensor(wt, dtype=torch.float32, device=dev), cin, use_workspace=False)
    assert np.abs(one.cpu().numpy() - want).max() <= 2e-4 * scale         # the one-kernel form (no workspace)


@pytest.mark.parametrize('n,h,w,k,stride,cin,cout,wino', [
    (3, 20, 28, 1, 1, 128, 64, 0),      # 128x64 tiles
    (3, 20, 28, 1, 1, 256, 128, 0),     # 64x64 tiles (1x1, Cout > 64)
    (2, 26, 26, 3, 2, 64, 128, 0),      # stride 2, data-parallel 128x128
    (32, 52, 52, 3, 2, 128, 256, 0),    # stride 2 on the stream-K schedule (cut tiles summed in the kernel)
    (2, 24, 24, 1, 1, 64, 32, 0),       # 128x32 tiles
    (3, 13, 13, 3, 1, 64, 128, 1),      # Winograd, odd map (outputs that do not exist must not be counted)
    (16, 52, 52, 3, 1, 128, 256, 1),    # Winograd on the stream-K schedule
    (2, 26, 26, 3, 1, 64, 64, 0),       # direct 3x3 stride 1
    (3, 13, 13, 3, 1, 64, 128, 2),      # F(4x4,3x3), odd map: 4x4 tiles hang over the edge (not counted), ragged last block
    (8, 52, 52, 3, 1, 128, 256, 2),     # F(4x4,3x3), several blocks per image
    (6, 13, 13, 3, 1, 64, 128, 2),      # F(4x4,3x3) on an odd map: the batch tiled as one mosaic (tiles straddle images)
    (6, 13, 13, 3, 1, 64, 128, 3),      # ... and the one-kernel form of the same (no workspace)
    (8, 52, 52, 3, 1, 128, 256, 3),
])
def test_conv_epilogue_statistics_equal_the_separate_pass(n, h, w, k, stride, cin, cout, wino):
    """Training forward (ref: model.py:35-41 with is_training=True): the conv writes per-row-block column sums of its
    output in the epilogue and y3_bn_train_stats_partials finishes them; mean / inv_std / scale / shift / moving statistics
    must agree with y3_bn_train_stats on the same z (1e-6 relative: only the fp32 partial-sum grouping differs), the conv
    output must be bit-identical to the plain entry point, and everything is run-to-run bit-exact."""
    from yolov3_tensorflow_amd import engine
    fw, _lib, L, ctx = _ctx()
    dev = fw.default_device()
    rng = np.random.RandomState(cin + cout + k + stride)
    x = torch.tensor(rng.standard_normal((n, h, w, cin)).astype(np.float32), device=dev)
    wt = torch.tensor((rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32), device=dev)
    ones, zeros = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    gamma = torch.tensor(rng.uniform(0.5, 1.5, cout).astype(np.float32), device=dev)
    beta = torch.tensor(rng.normal(0, 0.3, cout).astype(np.float32), device=dev)
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, k, stride, 0)
    nblk = L.y3_conv_stats_blocks(ctypes.byref(d), min(wino, 2))
    assert nblk > 0
    if wino >= 2:                        # 2: the two-kernel form (workspace), 3: the one-kernel form
        wp = engine.pack_wino44(wt)
        conv = lambda stats: engine.conv2d_fwd_wino44(x, wp, ones, zeros, cout, False, use_workspace=wino == 2, stats=stats)
    elif wino:
        wp = engine.pack_wino(wt)
        conv = lambda stats: engine.conv2d_fwd_wino(x, wp, ones, zeros, cout, False, stats=stats)
    else:
        wp = torch.empty(k * k * cout * cin, device=dev)
        _lib.check(L.y3_pack_conv_weights(ctx, fw.ptr(wt), k, cin, cout, fw.ptr(wp)))
        conv = lambda stats: engine.conv2d_fwd(x, wp, ones, zeros, k, stride, cout, False, stats=stats)
    z_plain = conv(None)
    outs = []
    for _ in range(2):
        part = torch.full((nblk, 2, cout), float('nan'), device=dev)
        z = conv(part)
        assert torch.equal(z, z_plain)
        st = torch.empty((4, cout), device=dev)
        mm, mv = torch.full((cout,), 0.25, device=dev), torch.full((cout,), 2.0, device=dev)
        rows = z.numel() // cout
        _lib.check(L.y3_bn_train_stats_partials(ctx, fw.ptr(part), nblk, rows, cout, fw.ptr(gamma), fw.ptr(beta),
                                                ctypes.c_float(1e-5), ctypes.c_float(0.9), fw.ptr(st[0]), fw.ptr(st[1]),
                                                fw.ptr(st[2]), fw.ptr(st[3]), fw.ptr(mm), fw.ptr(mv)))
        outs.append((part, st, mm, mv))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b), 'not run-to-run bit-exact'
    st2 = torch.empty((4, cout), device=dev)
    mm2, mv2 = torch.full((cout,), 0.25, device=dev), torch.full((cout,), 2.0, device=dev)
    sc = torch.empty(L.y3_reduce_scratch_bytes(cout), dtype=torch.uint8, device=dev)
    _lib.check(L.y3_bn_train_stats(ctx, fw.ptr(z_plain), z_plain.numel() // cout, cout, fw.ptr(gamma), fw.ptr(beta),
                                   ctypes.c_float(1e-5), ctypes.c_float(0.9), fw.ptr(st2[0]), fw.ptr(st2[1]),
                                   fw.ptr(st2[2]), fw.ptr(st2[3]), fw.ptr(mm2), fw.ptr(mv2), fw.ptr(sc)))
    _, st, mm, mv = outs[0]
    zz = z_plain.double().reshape(-1, cout)
    mean64, var64 = zz.mean(0), zz.var(0, unbiased=False)
    assert float((st[0].double() - mean64).abs().max()) <= 1e-5 * float(zz.abs().max())
    np.testing.assert_allclose(st[1].cpu().numpy(), (1.0 / torch.sqrt(var64 + 1e-5)).cpu().numpy(), rtol=2e-5)
    for a, b in ((st, st2), (mm, mm2), (mv, mv2)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('rows,c', [(2 * 13 * 13, 1024), (3 * 20 * 28, 64), (5000, 32), (64, 256)])
def test_bn_train_forward_backward(rows, c):
    fw, _lib, L, ctx = _ctx()
    dev = fw.default_device()
    rng = np.random.RandomState(rows + c)
    z = torch.tensor(rng.standard_normal((rows, c)) * 2 + rng.standard_normal(c), dtype=torch.float64, requires_grad=True)
    gamma = torch.tensor(rng.uniform(0.5, 1.5, c), dtype=torch.float64, requires_grad=True)
    beta = torch.tensor(rng.normal(0, 0.3, c), dtype=torch.float64, requires_grad=True)
    resid = rng.standard_normal((rows, c))
    mean, var = z.mean(0), z.var(0, unbiased=False)
    u = (z - mean) * gamma / torch.sqrt(var + 1e-5) + beta
    y = torch.where(u > 0, u, 0.1 * u) + torch.tensor(resid)
    dy = rng.standard_normal((rows, c))
    y.backward(torch.tensor(dy))
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    zg, gg, bg = f32(z.detach().numpy()), f32(gamma.detach().numpy()), f32(beta.detach().numpy())
    stats = torch.empty((4, c), device=dev)
    mm, mv = f32(np.full(c, 0.25)), f32(np.full(c, 2.0))
    sc = torch.empty(L.y3_bn_bwd_scratch_bytes(c), dtype=torch.uint8, device=dev)
    _lib.check(L.y3_bn_train_stats(ctx, fw.ptr(zg), rows, c, fw.ptr(gg), fw.ptr(bg), ctypes.c_float(1e-5),
                                   ctypes.c_float(0.9), fw.ptr(stats[0]), fw.ptr(stats[1]), fw.ptr(stats[2]),
                                   fw.ptr(stats[3]), fw.ptr(mm), fw.ptr(mv), fw.ptr(sc)))
    yg = torch.empty((rows, c), device=dev)
    _lib.check(L.y3_bn_apply_fwd(ctx, fw.ptr(zg), fw.ptr(stats[2]), fw.ptr(stats[3]), fw.ptr(f32(resid)), rows, c, 1,
                                 fw.ptr(yg)))
    assert rel_err(stats[0].cpu().numpy(), mean.detach().numpy()) < 1e-5
    assert rel_err(yg.cpu().numpy(), y.detach().numpy()) < 2e-5
    unb = var.detach().numpy() * rows / (rows - 1.0)
    np.testing.assert_allclose(mm.cpu().numpy(), 0.25 * 0.9 + mean.detach().numpy() * 0.1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mv.cpu().numpy(), 2.0 * 0.9 + unb * 0.1, rtol=1e-5, atol=1e-6)
    dyg = f32(dy)
    dgam, dbet = torch.empty(c, device=dev), torch.empty(c, device=dev)
    _lib.check(L.y3_bn_train_bwd(ctx, fw.ptr(zg), fw.ptr(dyg), fw.ptr(gg), fw.ptr(stats[2]), fw.ptr(stats[3]),
                                 fw.ptr(stats[0]), fw.ptr(stats[1]), rows, c, fw.ptr(dgam), fw.ptr(dbet),
                                 fw.ptr(dyg), fw.ptr(sc)))
    assert rel_err(dgam.cpu().numpy(), gamma.grad.numpy()) < 2e-4
    assert rel_err(dbet.cpu().numpy(), beta.grad.numpy()) < 2e-4
    assert rel_err(dyg.cpu().numpy(), z.grad.numpy()) < 2e-4


@pytest.mark.parametrize('smooth,focal', [(False, False), (True, True)])
def test_loss_and_its_gradient_match_oracle(smooth, focal):
    import yolov3_tensorflow_amd as y3
    from oracle import train_ref
    rng = np.random.RandomState(3)
    n, size = 3, 160
    model = y3.yolov3(80, COCO_ANCHORS, use_label_smooth=smooth, use_focal_loss=focal)
    model.img_size = [size, size]
    fms = [(rng.standard_normal((n, size // s, size // s, 255)) * 1.5).astype(np.float32) for s in (32, 16, 8)]
    yts = train_ref.synthetic_targets(9, n, [size, size], 80, COCO_ANCHORS, max_boxes=6)
    yts[1][1] = 0; yts[1][1][..., -1] = 1          # one image without objects on the 26-grid scale
    g = train_ref.TrainGraph({}, 80, torch.float64)
    g.img_size = [size, size]
    tf = [torch.tensor(f, dtype=torch.float64, requires_grad=True) for f in fms]
    ref = g.compute_loss(tf, yts, COCO_ANCHORS, smooth, focal)
    ref[0].backward()
    got = model.compute_loss(fms, yts)
    for a, b in zip(got, ref):
        assert abs(float(a) - float(b)) <= 1e-4 * abs(float(b)) + 1e-6, (float(a), float(b))
    for gg, t in zip(model._train['fm_grads'], tf):
        gnp = gg.cpu().numpy()
        assert np.all(gnp[..., 255:] == 0)
        assert rel_err(gnp[..., :255], t.grad.numpy()) < 2e-4
    # the single-scale API of the reference
    parts = model.loss_layer(fms[0], yts[0], COCO_ANCHORS[6:9])
    ref0 = g.loss_layer(torch.tensor(fms[0], dtype=torch.float64), yts[0], COCO_ANCHORS[6:9], smooth, focal)
    for a, b in zip(parts, ref0):
        assert abs(float(a) - float(b)) <= 1e-4 * abs(float(b)) + 1e-6
    with pytest.raises(ValueError):
        model.compute_loss(fms, [yts[0], yts[0], yts[2]])


@pytest.mark.parametrize('kind', [0, 1, 2, 3])
def test_multi_tensor_clip_update_equals_the_per_tensor_form(kind):
    """y3_clip_update_multi (3 launches for all tensors; ref: train.py:112-115) against y3_clip_update tensor by tensor:
    same arithmetic, only the association of the norm's partial sums differs (-> 1e-6 relative), deterministic."""
    fw, _lib, L, ctx = _ctx()
    dev = fw.default_device()
    rng = np.random.RandomState(kind)
    sizes = [255, 32, 3 * 3 * 64 * 128, 1024, 18, 8192, 8193, 3 * 3 * 256 * 512, 7]
    def make():
        out = []
        r = np.random.RandomState(100 + kind)
        for i, n in enumerate(sizes):
            big = 50.0 if i % 3 == 0 else 0.01          # some tensors get clipped (norm > 100), some do not
            t = lambda a: torch.tensor(a.astype(np.float32), device=dev)
            out.append(dict(w=t(r.standard_normal(n)), g=t(r.standard_normal(n) * big), s0=t(np.abs(r.standard_normal(n))),
                            s1=t(np.abs(r.standard_normal(n))), wd=5e-4 if n > 1000 else 0.0))
        return out
    hp = dict(gs=0.5, clip=100.0, lr=1e-2, mom=0.9, decay=0.9, b2=0.999, eps=1e-8)
    def run_multi(ts):
        arr = (_lib.ParamDesc * len(ts))()
        for i, t in enumerate(ts):
            arr[i] = _lib.ParamDesc(t['w'].data_ptr(), t['g'].data_ptr(), t['s0'].data_ptr(), t['s1'].data_ptr(),
                                    t['w'].numel(), t['wd'], 0)
        sc = torch.empty(L.y3_clip_update_multi_scratch_bytes(arr, len(ts)), dtype=torch.uint8, device=dev)
        _lib.check(L.y3_clip_update_multi(ctx, kind, arr, len(ts), ctypes.c_float(hp['gs']), ctypes.c_float(hp['clip']),
                                          ctypes.c_float(hp['lr']), ctypes.c_float(hp['mom']), ctypes.c_float(hp['decay']),
                                          ctypes.c_float(hp['b2']), ctypes.c_float(hp['eps']), fw.ptr(sc),
                                          ctypes.c_size_t(sc.numel())))
    a, b, c = make(), make(), make()
    run_multi(a)
    run_multi(c)
    sc = torch.empty(L.y3_optimizer_scratch_bytes(), dtype=torch.uint8, device=dev)
    for t in b:
        _lib.check(L.y3_clip_update(ctx, kind, fw.ptr(t['w']), fw.ptr(t['g']), fw.ptr(t['s0']), fw.ptr(t['s1']),
                                    t['w'].numel(), ctypes.c_float(t['wd']), ctypes.c_float(hp['gs']),
                                    ctypes.c_float(hp['clip']), ctypes.c_float(hp['lr']), ctypes.c_float(hp['mom']),
                                    ctypes.c_float(hp['decay']), ctypes.c_float(hp['b2']), ctypes.c_float(hp['eps']),
                                    fw.ptr(sc)))
    for ta, tb, tc in zip(a, b, c):
        for key in ('w', 'g', 's0', 's1'):
            assert torch.equal(ta[key], tc[key]), 'multi-tensor update is not deterministic'
            np.testing.assert_allclose(ta[key].cpu().numpy(), tb[key].cpu().numpy(), rtol=2e-6, atol=1e-7,
                                       err_msg='%s of a %d-element tensor' % (key, ta['w'].numel()))
    # clipping happened where it should: ||g|| == 100 for the large-gradient tensors
    assert abs(float(a[0]['g'].norm()) - 100.0) < 1e-2 and float(a[1]['g'].norm()) < 100.0


def _fresh_model(params, **kw):
    import yolov3_tensorflow_amd as y3
    y3.reset_default_graph()
    model = y3.yolov3(80, COCO_ANCHORS, **kw)
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 32, 32, 3))
    for v in y3.global_variables(scope='yolov3'):
        v.assign(params[v.op_name])
    return model


_REF_CACHE = {}      # (dtype, update_scopes) -> (mask checksum, fp64 oracle step with sgd)


def _conv_name(i):
    sub, j = ('darknet53_body', i) if i < 52 else ('yolov3_head', i - 52)
    return 'yolov3/%s/%s' % (sub, 'Conv' if j == 0 else 'Conv_%d' % j)


@pytest.mark.parametrize('optimizer,update_scopes,dtype', [
    ('sgd', None, 'f32'), ('momentum', None, 'f32'), ('adam', None, 'f32'), ('rmsprop', None, 'f32'),
    ('momentum', ['yolov3/yolov3_head'], 'f32'),
    # forward + stride-1 data gradients on the bf16 matrix pipe: same oracle, same tolerances
    ('sgd', None, 'f32_bf16x6'), ('adam', None, 'f32_bf16x6'),
    # Winograd forward for the stride-1 3x3 convs (backward unchanged)
    ('sgd', None, 'f32_wino')])
def test_one_train_step_matches_oracle(optimizer, update_scopes, dtype, isolated_graph):
    """One whole train step (ref: train.py:105-115) at 256 px, bs=4 (every BN layer reduces over >= 256 samples) against
    the fp64 autograd oracle: loss 5-tuple 1e-4, EVERY clipped gradient tensor within 2e-4 of its max magnitude,
    updated variables and BN moving statistics.

    Conditioning: LeakyReLU makes the gradient discontinuous in the forward values — an element whose pre-activation
    changes sign between two implementations flips a 1/0.1 factor, and a fraction p of flipped elements moves a
    gradient tensor by ~sqrt(p).  Measured on this very configuration: the CPU fp32 oracle with its OWN branches is
    4.4e-2 (worst tensor; median 3e-3) from the fp64 oracle, and 2.2e-5 when the activation is smooth.  So the oracle
    is run with the branches the GPU took (mask = z*scale+shift > 0 from the tensors the GPU kept for its backward):
    both sides then differentiate the same piecewise-linear function and every tensor is held to 2e-4.  The
    comparison against the oracle's own branches is printed for scale."""
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from oracle import yolo_ref, train_ref
    params = yolo_ref.synthetic_params(80, seed=1)
    n, size = 4, 256
    x = blob_images(21, n, size)
    yts = train_ref.synthetic_targets(5, n, [size, size], 80, COCO_ANCHORS, max_boxes=4)
    lr = 1e-3
    model = _fresh_model(params, batch_norm_decay=0.99, weight_decay=5e-4)
    model.compute_dtype = dtype
    upd = None if update_scopes is None else [v for v in y3.global_variables(scope='yolov3')
                                              if any(v.op_name.startswith(s) for s in update_scopes)]
    trainer = training.Trainer(model, config_optimizer(optimizer, lr), update_vars=upd)
    trainer.capture = []
    with y3.variable_scope('yolov3'):
        loss = trainer.step(x, yts)
    # the LeakyReLU branches the GPU took, per BN layer (captured layers: those backward visited)
    masks, csum = {}, 0
    for rec in trainer.capture:
        if rec['z'] is None:
            continue
        pos = (rec['z'] * rec['stats'][2] + rec['stats'][3]) > 0
        masks[_conv_name(rec['layer'])] = pos.permute(0, 3, 1, 2).cpu()
        csum += int(pos.sum().item()) * (rec['layer'] + 1)
    trainer.capture = None
    key = (dtype, None if update_scopes is None else tuple(update_scopes))
    if key not in _REF_CACHE or _REF_CACHE[key][0] != csum:
        _REF_CACHE[key] = (csum, train_ref.train_step(params, x, yts, COCO_ANCHORS, optimizer='sgd', lr=lr,
                                                      weight_decay=5e-4, bn_decay=0.99, update_scopes=update_scopes,
                                                      dtype=torch.float64, step=1, masks=masks))
    ref = dict(_REF_CACHE[key][1])
    ref['new_params'] = train_ref.reapply(params, ref, optimizer, lr, step=1)
    for a, b in zip(loss, ref['loss']):
        assert abs(float(a) - b) <= 1e-4 * abs(b) + 1e-6, ([float(v) for v in loss], ref['loss'])
    # clipped gradients (incl. the L2 term), every trainable variable
    assert set(trainer.views) == set(ref['grads'])
    GRAD_TOL = 2e-4      # measured: worst 2.0e-5 over all tensors, modes and optimizers (vs 1.1e-1 on the oracle's own branches)
    errs = {name: rel_err(trainer.views[name].cpu().numpy(), g) for name, g in ref['grads'].items()}
    worst = max(errs, key=errs.get)
    msg = '%s/%s: gradient rel err vs fp64 oracle on the GPU\'s branches: worst %.2e (%s), median %.2e over %d tensors' % (
        optimizer, dtype, errs[worst], worst, float(np.median(list(errs.values()))), len(errs))
    if optimizer == 'sgd' and update_scopes is None:
        own = _REF_CACHE.get(('own', None))
        if own is None:
            own = train_ref.train_step(params, x, yts, COCO_ANCHORS, optimizer='sgd', lr=lr, weight_decay=5e-4,
                                       bn_decay=0.99, dtype=torch.float64, step=1)
            _REF_CACHE[('own', None)] = own
        e_own = [rel_err(trainer.views[k].cpu().numpy(), g) for k, g in own['grads'].items()]
        msg += ' ; vs the oracle\'s own branches: worst %.2e, median %.2e' % (max(e_own), float(np.median(e_own)))
    print(msg)
    for name, e in errs.items():
        assert e < GRAD_TOL, '%s: grad rel err %.3e' % (name, e)
    # updated variables and BN moving statistics
    for v in y3.global_variables(scope='yolov3'):
        want = ref['new_params'][v.op_name]
        got = v.numpy()
        scale = max(np.abs(want).max(), 1e-6)
        diff = np.abs(got - want)
        if optimizer in ('sgd', 'momentum') or v.op_name not in ref['grads']:
            slack = 0.0
            if v.op_name in ref['grads']:      # the gradient tolerance above, times the step
                slack = GRAD_TOL * lr * float(np.abs(ref['grads'][v.op_name]).max())
            assert diff.max() <= 1e-4 * scale + slack, v.op_name
        else:
            # adam / rmsprop normalise by sqrt(v): the first step is ~ lr*sign(g), so where the gradient is
            # numerically zero the sign (hence a 2*lr difference) is noise; elsewhere the step must agree
            g = np.abs(ref['grads'][v.op_name])
            solid = g > 1e-2 * g.max()
            assert diff.max() <= 2.2 * lr + 1e-6, v.op_name
            assert diff[solid].max() <= 5e-2 * lr + 1e-4 * scale, v.op_name
    if update_scopes is not None:
        body_w = 'yolov3/darknet53_body/Conv_5/weights'
        np.testing.assert_array_equal(dict((v.op_name, v) for v in y3.global_variables())[body_w].numpy(),
                                      params[body_w])


def test_train_step_is_deterministic_and_loss_decreases(isolated_graph):
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from oracle import yolo_ref, train_ref
    params = yolo_ref.synthetic_params(80, seed=2)
    x = blob_images(4, 2, 96)
    yts = train_ref.synthetic_targets(6, 2, [96, 96], 80, COCO_ANCHORS, max_boxes=3)
    runs = []
    for _ in range(2):
        model = _fresh_model(params, batch_norm_decay=0.99)
        trainer = training.Trainer(model, config_optimizer('momentum', 1e-3))
        with y3.variable_scope('yolov3'):
            losses = [float(trainer.step(x, yts)[0]) for _ in range(4)]
        runs.append((losses, trainer.flat.clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][-1] < runs[0][0][0]


@pytest.mark.parametrize('dtype', ['f32', 'f32_wino'])
def test_weight_gradients_on_the_second_stream_change_no_bit(dtype, isolated_graph):
    """y3_net_train_set_wgrad_stream (the default of training.Trainer): the same kernels on the same inputs, only on another
    stream - every loss, every gradient and the order of the `ready` calls must be those of the one-stream pass; the
    workspace holds one layer's dz one layer longer."""
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from oracle import yolo_ref, train_ref
    params = yolo_ref.synthetic_params(80, seed=4)
    x = blob_images(7, 3, 128)
    yts = train_ref.synthetic_targets(8, 3, [128, 128], 80, COCO_ANCHORS, max_boxes=4)
    runs = []
    for side in (False, True, torch.cuda.Stream()):          # off / the library's low-priority stream / a stream of the caller's
        model = _fresh_model(params, batch_norm_decay=0.99)
        model.compute_dtype = dtype
        trainer = training.Trainer(model, config_optimizer('momentum', 1e-3), wgrad_stream=side)
        edges = []
        with y3.variable_scope('yolov3'):
            losses = [float(trainer.step(x, yts)[0]) for _ in range(3)]
            ready = trainer.exchange.ready
            trainer.exchange.ready = lambda edge: (edges.append(int(edge)), ready(edge))[1]
            losses.append(float(trainer.step(x, yts)[0]))
        torch.cuda.synchronize()
        runs.append((losses, trainer.flat.clone(), edges, model._train['ws'].numel()))
    # ... and 'auto' (the default) measures both in its first seven steps and settles on one of them
    model = _fresh_model(params, batch_norm_decay=0.99)
    model.compute_dtype = dtype
    trainer = training.Trainer(model, config_optimizer('momentum', 1e-3))
    with y3.variable_scope('yolov3'):
        losses = [float(trainer.step(x, yts)[0]) for _ in range(8)]
    assert trainer.wgrad_choice in (True, False) and set(trainer.wgrad_calibration) == {'ms_with', 'ms_without'}
    assert losses[:4] == runs[0][0]
    for other in runs[1:]:
        assert other[0] == runs[0][0]
        assert torch.equal(other[1], runs[0][1])
        assert other[2] == runs[0][2] and len(other[2]) > 10
    assert runs[1][3] >= runs[0][3]


def test_train_workspace_follows_the_dtype_and_a_small_one_is_refused_before_launch(isolated_graph):
    """(a) the step's workspace is sized per (shape, device, compute_dtype) - the allocation sequence differs
    between 'f32', 'f32_wino' and 'f32_bf16x6' - so switching the mode on a live model must re-size it, and the steps
    stay finite and deterministic per mode; (b) y3_net_train_forward compares the workspace with the dry-run peak for
    the current dtype BEFORE it enqueues anything: a short workspace is Y3_EINVAL, nothing is written."""
    import ctypes
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training, _lib, framework as fw
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from oracle import yolo_ref, train_ref
    params = yolo_ref.synthetic_params(80, seed=4)
    x = blob_images(7, 2, 96)
    yts = train_ref.synthetic_targets(8, 2, [96, 96], 80, COCO_ANCHORS, max_boxes=3)
    model = _fresh_model(params, batch_norm_decay=0.99)
    trainer = training.Trainer(model, config_optimizer('sgd', 1e-4))
    sizes = {}
    with y3.variable_scope('yolov3'):
        for mode in ('f32', 'f32_wino', 'f32_bf16x6', 'f32'):
            model.compute_dtype = mode
            loss = trainer.step(x, yts)
            assert np.isfinite(float(loss[0])), mode
            st = training._train_state(model)
            assert st['ws_shape'][-1] == mode
            sizes.setdefault(mode, st['ws'].numel())
        fw.check_context()
        # (b) a workspace one byte-page short of the dry-run peak: refused, and the guard word behind it is untouched
        st, net, _ = training._prepare(model, fw.as_device_f32(x))
        L = _lib.lib()
        need = L.y3_net_train_workspace_bytes(net, st['vars_all'], 2, 96, 96)
        assert need > 4096
        ws = torch.zeros(need + 256, dtype=torch.uint8, device=fw.default_device())
        ws[need - 4096:] = 0xA5
        opts, _anchors_keepalive = training._opts(model)
    xt = fw.as_device_f32(x)
    rc = L.y3_net_train_forward(net, st['vars_all'], fw.ptr(xt), 2, 96, 96, ctypes.byref(opts), fw.ptr(ws),
                                ctypes.c_size_t(need - 4096), None, None, None)
    torch.cuda.synchronize()
    assert rc == _lib.Y3_EINVAL
    assert 'workspace too small' in L.y3_last_error().decode()
    assert bool((ws[need - 4096:] == 0xA5).all())


def test_box_iou_matches_oracle():
    import yolov3_tensorflow_amd as y3
    from oracle import train_ref
    rng = np.random.RandomState(8)
    pred = np.concatenate([rng.uniform(0, 416, (13, 13, 3, 2)), rng.uniform(5, 300, (13, 13, 3, 2))], -1).astype(np.float32)
    gt = np.concatenate([rng.uniform(0, 416, (7, 2)), rng.uniform(5, 300, (7, 2))], -1).astype(np.float32)
    model = y3.yolov3(80, COCO_ANCHORS)
    got = model.box_iou(pred, gt).cpu().numpy()
    want = train_ref.TrainGraph.box_iou(torch.tensor(pred, dtype=torch.float64), torch.tensor(gt, dtype=torch.float64)).numpy()
    assert got.shape == (13, 13, 3, 7)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
