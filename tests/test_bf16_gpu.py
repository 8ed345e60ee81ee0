"""GPU tests of the bf16-storage path (BASELINE configs[4]).  Per-op: the bf16 conv against an fp64 convolution
of the SAME bf16-rounded operands — the only differences are the fp32 accumulation order and the single final
rounding, so the bound is one bf16 ulp (2^-8 relative) plus fp32 accumulation noise.  Whole network (416 and the
configs[4] size 608): gated against the fp32 oracle at 3 % max / 2 % rms of the logits (bf16 is not held to the 1e-3
box tolerance)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import blob_images

pytestmark = pytest.mark.gpu


def bf16_round(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def ref_conv(x, w_hwio, scale, shift, k, stride, act, resid=None):
    xd = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wd = torch.from_numpy(w_hwio).double().permute(3, 2, 0, 1)
    if stride > 1:
        y = F.conv2d(F.pad(xd, (1, 1, 1, 1)), wd, stride=stride)
    else:
        y = F.conv2d(xd, wd, padding=k // 2)
    y = y * torch.from_numpy(scale).double().view(1, -1, 1, 1) + torch.from_numpy(shift).double().view(1, -1, 1, 1)
    if act:
        y = torch.where(y > 0, y, 0.1 * y)
    y = y.permute(0, 2, 3, 1)
    if resid is not None:
        y = y + torch.from_numpy(resid).double()
    return y.numpy()


@pytest.mark.parametrize('n,h,w,k,stride,cin,cout,resid,c_up,out_f32', [
    (3, 20, 28, 3, 1, 128, 256, True, 0, 0), (2, 26, 26, 3, 2, 64, 128, False, 0, 0),
    (3, 20, 28, 1, 1, 256, 128, False, 0, 0), (2, 26, 26, 1, 1, 768, 256, False, 256, 0),
    (2, 19, 19, 1, 1, 1024, 255, False, 0, 1), (3, 20, 28, 3, 1, 32, 64, True, 0, 0),
    (2, 24, 24, 1, 1, 64, 32, False, 0, 0), (1, 13, 13, 3, 1, 512, 1024, False, 0, 0),
])
def test_bf16_conv_matches_fp64_on_rounded_operands(n, h, w, k, stride, cin, cout, resid, c_up, out_f32):
    from yolov3_tensorflow_amd import framework as fw, _lib
    dev = fw.default_device()
    L, ctx = _lib.lib(), fw.context()
    rng = np.random.RandomState(cin * 7 + cout)
    cx = cin - c_up
    x = bf16_round(rng.standard_normal((n, h, w, cx)))
    xu = bf16_round(rng.standard_normal((n, h // 2, w // 2, c_up))) if c_up else None
    wt = bf16_round(rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin)))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.2, cout).astype(np.float32)
    r = bf16_round(rng.standard_normal((n, h // stride, w // stride, cout))) if resid else None
    xin = x if xu is None else np.concatenate([np.repeat(np.repeat(xu, 2, 1), 2, 2), x], axis=3)
    want = ref_conv(xin, wt, scale, shift, k, stride, not out_f32, r)
    tb = lambda a: None if a is None else torch.from_numpy(a).to(dev).to(torch.bfloat16).contiguous()
    wg = torch.from_numpy(wt).to(dev)
    wp = torch.empty(k * k * cout * cin, dtype=torch.bfloat16, device=dev)
    _lib.check(L.y3_pack_conv_weights_bf16(ctx, fw.ptr(wg), k, cin, cout, fw.ptr(wp)))
    y = torch.empty((n, h // stride, w // stride, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    d = _lib.ConvDesc(n, h, w, cin, c_up, cout, k, stride, 0 if out_f32 else 1)
    xg, xug, rg = tb(x), tb(xu), tb(r)                     # keep the device buffers alive across the launch
    scg, shg = torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)
    _lib.check(L.y3_conv2d_fwd_bf16(ctx, ctypes.byref(d), fw.ptr(xg), fw.ptr(xug), fw.ptr(wp), fw.ptr(scg),
                                    fw.ptr(shg), fw.ptr(rg), fw.ptr(y), out_f32))
    got = y.float().cpu().numpy()
    tol = (1e-4 if out_f32 else 2.0 ** -8) * np.abs(want) + 2e-3
    assert np.isfinite(got).all()
    assert (np.abs(got - want) <= tol).all(), float(np.abs(got - want).max())


@pytest.mark.parametrize('size', [416, 608])
def test_bf16_forward_tracks_the_fp32_oracle(gpu_model, size):
    """configs[4] (608x608 bf16 storage) and the 416 size: the deviation from the fp32 oracle is GATED, not just
    reported — max |d| <= 3 % of the largest logit, rms <= 2 % (measured: 1.6 % / 1.2 %: 75 layers of bf16 rounding at
    2^-9 relative each); bf16 is not held to the 1e-3 box tolerance of the fp32 path."""
    import yolov3_tensorflow_amd as y3
    from oracle import yolo_ref
    model, params = gpu_model
    x = blob_images(0, 2 if size == 416 else 1, size)
    ref = yolo_ref.forward(params, x)
    model.compute_dtype = 'bf16'
    try:
        with y3.variable_scope('yolov3'):
            fms = model.forward(x, False)
            again = model.forward(x, False)
    finally:
        model.compute_dtype = 'f32'
    for i, (g, g2, r) in enumerate(zip(fms, again, ref)):
        assert g.dtype == torch.float32 and torch.equal(g, g2)
        g = g.cpu().numpy()
        rel = float(np.abs(g - r).max() / np.abs(r).max())
        rms = float(np.sqrt(((g - r) ** 2).mean()) / np.sqrt((r ** 2).mean()))
        print('bf16 feature_map_%d: max err / max|ref| = %.3e, rms rel = %.3e' % (i + 1, rel, rms))
        assert np.isfinite(g).all() and rel < 0.03 and rms < 0.02
    # fp32 path is untouched by the excursion
    with y3.variable_scope('yolov3'):
        f32 = model.forward(x, False)
    assert np.abs(f32[0].cpu().numpy() - ref[0]).max() < 2e-4
