"""GPU parity: y3_conv2d_fwd (through the C ABI) against an fp64 torch-CPU convolution of the same op, on
every distinct conv shape of the network at reduced spatial size, plus true-size cases.
Tolerance (stated here, fp32 accumulate over K <= 4608): |d| <= 1e-4 + 1e-4*|ref|.
The same cases run through y3_conv2d_fwd_split with planes = 3 (fp32 products rebuilt from 6 bf16 plane products)
at the SAME tolerance, and with planes = 2 (3 products, dropped terms 2^-15) at 2e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (k, stride, cin, cout, bn(act), residual)
DISTINCT = [
    (3, 1, 3, 32, True, False), (3, 2, 32, 64, True, False), (1, 1, 64, 32, True, False),
    (3, 1, 32, 64, True, True), (3, 2, 64, 128, True, False), (1, 1, 128, 64, True, False),
    (3, 1, 64, 128, True, True), (3, 2, 128, 256, True, False), (1, 1, 256, 128, True, False),
    (3, 1, 128, 256, True, True), (3, 2, 256, 512, True, False), (1, 1, 512, 256, True, False),
    (3, 1, 256, 512, True, True), (3, 2, 512, 1024, True, False), (1, 1, 1024, 512, True, False),
    (3, 1, 512, 1024, True, True), (3, 1, 512, 1024, True, False), (1, 1, 1024, 255, False, False),
    (1, 1, 512, 255, False, False), (1, 1, 256, 255, False, False), (3, 1, 256, 512, True, False),
    (3, 1, 128, 256, True, False), (1, 1, 1024, 18, False, False),
]


def ref_conv(x, w_hwio, scale, shift, k, stride, act, resid=None):
    """fp64 restatement: explicit pad for stride 2 (utils/layer_utils.py:10-21), SAME for stride 1."""
    xd = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wd = torch.from_numpy(w_hwio).double().permute(3, 2, 0, 1)
    if stride > 1:
        p = (k - 1) // 2
        xd = F.pad(xd, (p, k - 1 - p, p, k - 1 - p))
        y = F.conv2d(xd, wd, stride=stride)
    else:
        y = F.conv2d(xd, wd, padding=k // 2)
    y = y * torch.from_numpy(scale).double().view(1, -1, 1, 1) + torch.from_numpy(shift).double().view(1, -1, 1, 1)
    if act:
        y = torch.where(y > 0, y, 0.1 * y)
    y = y.permute(0, 2, 3, 1)
    if resid is not None:
        y = y + torch.from_numpy(resid).double()
    return y.numpy()


# 0: exact fp32 kernel; 3 / 2: y3_conv2d_fwd_split (fp32 tensors, operands split into 3 / 2 bf16 planes)
PLANES = [0, 3, 2]
TOL = {0: 1e-4, 3: 1e-4, 2: 2e-3}


def run_gpu(x, w_hwio, scale, shift, k, stride, act, resid=None, x_up=None, planes=0):
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    L = _lib.lib()
    cin_total = w_hwio.shape[2]
    cout = w_hwio.shape[3]
    w = torch.from_numpy(w_hwio).to(dev)
    if cin_total == 3:
        wp = w
    elif planes > 0:
        wp = torch.empty(planes * k * k * cout * cin_total, device=dev, dtype=torch.bfloat16)
        _lib.check(L.y3_pack_conv_weights_split(fw.context(), fw.ptr(w), k, cin_total, cout, planes, fw.ptr(wp)))
    elif planes == 0:
        wp = torch.empty(k * k * cout * cin_total, device=dev)
        _lib.check(L.y3_pack_conv_weights(fw.context(), fw.ptr(w), k, cin_total, cout, fw.ptr(wp)))
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    y = engine.conv2d_fwd(t(x), wp, t(scale), t(shift), k, stride, cout, act, residual=t(resid), x_up=t(x_up),
                          planes=planes)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def make_case(rng, n, h, w, k, cin, cout):
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.2, cout).astype(np.float32)
    return x, wt, scale, shift


def check(got, want, what, planes=0):
    err = np.abs(got - want)
    tol = TOL[planes] * (1 + np.abs(want))
    assert got.shape == want.shape, what
    assert np.isfinite(got).all(), what
    assert (err <= tol).all(), '%s: max err %.3e (max |ref| %.2f)' % (what, err.max(), np.abs(want).max())


@pytest.mark.parametrize('planes', PLANES)
@pytest.mark.parametrize('k,stride,cin,cout,act,resid', DISTINCT)
def test_distinct_shapes_reduced_spatial(k, stride, cin, cout, act, resid, planes):
    rng = np.random.RandomState(hash((k, stride, cin, cout, resid)) % (2 ** 31))
    n, h, w = 3, 20, 28                       # M = 1680 or 420: ragged last tile, non-square map
    x, wt, scale, shift = make_case(rng, n, h, w, k, cin, cout)
    r = rng.standard_normal((n, h // stride, w // stride, cout)).astype(np.float32) if resid else None
    got = run_gpu(x, wt, scale, shift, k, stride, act, r, planes=planes)
    check(got, ref_conv(x, wt, scale, shift, k, stride, act, r), 'k%d s%d %d->%d' % (k, stride, cin, cout), planes)


@pytest.mark.parametrize('planes', PLANES)
@pytest.mark.parametrize('c_up,c_route,cout', [(256, 512, 256), (128, 256, 128), (32, 32, 64)])
def test_fused_upsample_concat_input(c_up, c_route, cout, planes):
    rng = np.random.RandomState(c_up)
    n, h, w = 2, 26, 26
    xu = rng.standard_normal((n, h // 2, w // 2, c_up)).astype(np.float32)
    xr = rng.standard_normal((n, h, w, c_route)).astype(np.float32)
    _, wt, scale, shift = make_case(rng, 1, 1, 1, 1, c_up + c_route, cout)
    got = run_gpu(xr, wt, scale, shift, 1, 1, True, None, x_up=xu, planes=planes)
    cat = np.concatenate([np.repeat(np.repeat(xu, 2, 1), 2, 2), xr], axis=3)   # upsampled channels first
    check(got, ref_conv(cat, wt, scale, shift, 1, 1, True), 'upcat %d+%d' % (c_up, c_route), planes)


@pytest.mark.parametrize('n,h,w,k,stride,cin,cout', [
    (2, 52, 52, 3, 1, 128, 256),     # true-size 52x52 residual-stage conv
    (1, 416, 416, 3, 1, 3, 32),      # stem at full resolution
    (2, 104, 104, 3, 2, 128, 256),   # true-size downsample
    (1, 13, 13, 3, 1, 512, 1024),    # M = 169: two ragged M tiles, K = 4608
    (1, 19, 19, 1, 1, 1024, 255),    # 608-input head, odd map
    (5, 2, 2, 3, 1, 64, 64),         # map smaller than the kernel footprint
])
@pytest.mark.parametrize('planes', PLANES)
def test_true_size_and_edge_cases(n, h, w, k, stride, cin, cout, planes):
    rng = np.random.RandomState(n * 1000 + h)
    x, wt, scale, shift = make_case(rng, n, h, w, k, cin, cout)
    got = run_gpu(x, wt, scale, shift, k, stride, True, planes=planes)
    check(got, ref_conv(x, wt, scale, shift, k, stride, True),
          '%dx%dx%d k%d s%d %d->%d' % (n, h, w, k, stride, cin, cout), planes)


def test_linearity_and_zero_input_at_full_batch():
    """Size-independent properties at BASELINE size (bs=32, 52x52x128->256): conv(0) == shift exactly,
    conv(a*x) == a*conv(x) for a power of two (exact in fp32) with scale=1, shift=0, linear act."""
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    rng = np.random.RandomState(0)
    n, h, w, cin, cout, k = 32, 52, 52, 128, 256, 3
    wt = torch.from_numpy((rng.standard_normal((k, k, cin, cout)) * 0.03).astype(np.float32)).to(dev)
    wp = torch.empty(k * k * cout * cin, device=dev)
    _lib.check(_lib.lib().y3_pack_conv_weights(fw.context(), fw.ptr(wt), k, cin, cout, fw.ptr(wp)))
    ones, zeros = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    x0 = torch.zeros((n, h, w, cin), device=dev)
    y0 = engine.conv2d_fwd(x0, wp, ones, shift, k, 1, cout, False)
    assert torch.equal(y0, shift.view(1, 1, 1, -1).expand_as(y0))
    x = torch.randn((n, h, w, cin), device=dev)
    y1 = engine.conv2d_fwd(x, wp, ones, zeros, k, 1, cout, False)
    y4 = engine.conv2d_fwd(x * 4.0, wp, ones, zeros, k, 1, cout, False)
    assert torch.equal(y4, y1 * 4.0)
    # batch independence: bit-exact under the data-parallel schedule (fixed K order per output element);
    # under stream-K the split points of the K sum depend on M, so only to fp32 rounding
    y7 = engine.conv2d_fwd(x[7:8].contiguous(), wp, ones, zeros, k, 1, cout, False, use_workspace=False)
    y1d = engine.conv2d_fwd(x, wp, ones, zeros, k, 1, cout, False, use_workspace=False)
    assert torch.equal(y7[0], y1d[7])
    assert torch.allclose(y1, y1d, rtol=1e-5, atol=1e-5)
    # run-to-run determinism of the stream-K schedule
    assert torch.equal(engine.conv2d_fwd(x, wp, ones, zeros, k, 1, cout, False), y1)


@pytest.mark.parametrize('planes', [3, 2])
def test_split_full_batch_properties(planes):
    """planes=3/2 at BASELINE size (bs=32, 52x52x128->256, stream-K schedule): conv(0) == shift, exact
    power-of-two linearity (the bf16 planes scale exactly), run-to-run determinism, stream-K vs data-parallel
    agreement, and closeness to the exact-fp32 kernel (same tolerance as against the fp64 reference)."""
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    rng = np.random.RandomState(1)
    n, h, w, cin, cout, k = 32, 52, 52, 128, 256, 3
    wt = torch.from_numpy((rng.standard_normal((k, k, cin, cout)) * 0.03).astype(np.float32)).to(dev)
    wp = torch.empty(k * k * cout * cin, device=dev)
    _lib.check(_lib.lib().y3_pack_conv_weights(fw.context(), fw.ptr(wt), k, cin, cout, fw.ptr(wp)))
    ws = torch.empty(planes * k * k * cout * cin, device=dev, dtype=torch.bfloat16)
    _lib.check(_lib.lib().y3_pack_conv_weights_split(fw.context(), fw.ptr(wt), k, cin, cout, planes, fw.ptr(ws)))
    ones, zeros = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    x0 = torch.zeros((n, h, w, cin), device=dev)
    y0 = engine.conv2d_fwd(x0, ws, ones, shift, k, 1, cout, False, planes=planes)
    assert torch.equal(y0, shift.view(1, 1, 1, -1).expand_as(y0))
    x = torch.randn((n, h, w, cin), device=dev)
    y1 = engine.conv2d_fwd(x, ws, ones, zeros, k, 1, cout, False, planes=planes)
    y4 = engine.conv2d_fwd(x * 4.0, ws, ones, zeros, k, 1, cout, False, planes=planes)
    assert torch.equal(y4, y1 * 4.0)
    assert torch.equal(engine.conv2d_fwd(x, ws, ones, zeros, k, 1, cout, False, planes=planes), y1)
    y1d = engine.conv2d_fwd(x, ws, ones, zeros, k, 1, cout, False, use_workspace=False, planes=planes)
    assert torch.allclose(y1, y1d, rtol=1e-5, atol=1e-5)
    exact = engine.conv2d_fwd(x, wp, ones, zeros, k, 1, cout, False)
    err = (y1 - exact).abs()
    assert bool((err <= TOL[planes] * (1 + exact.abs())).all()), float(err.max())


def test_bad_arguments_raise_value_error():
    from yolov3_tensorflow_amd import engine, framework as fw
    dev = fw.default_device()
    x = torch.zeros((1, 8, 8, 24), device=dev)       # Cin not a multiple of 32
    w = torch.zeros(24 * 32, device=dev)
    s = torch.zeros(32, device=dev)
    with pytest.raises(ValueError):
        engine.conv2d_fwd(x, w, s, s, 1, 1, 32, True)
    x = torch.zeros((1, 7, 7, 32), device=dev)       # odd map with stride 2
    with pytest.raises(ValueError):
        engine.conv2d_fwd(x, torch.zeros(9 * 32 * 32, device=dev), s, s, 3, 2, 32, True)
    with pytest.raises(ValueError):
        engine.conv2d_fwd(x, torch.zeros(25 * 32 * 32, device=dev), s, s, 5, 1, 32, True)


WINO_CASES = [
    # n, h, w, cin, cout, act, resid
    (3, 20, 28, 64, 128, True, True),      # even map, ragged last tile block
    (2, 13, 13, 512, 1024, True, True),    # odd map (7x7 tiles, last tile half outside), deepest layer shape
    (1, 26, 26, 256, 512, True, False),
    (2, 52, 52, 128, 256, True, True),     # true-size 52x52 residual-stage conv
    (7, 26, 26, 64, 1024, True, True),     # 19 x 16 = 304 blocks on 256 workers: stream-K schedule, every block split
    (32, 13, 13, 64, 1024, True, False),   # BASELINE-size 13x13 map: 25 x 16 = 400 blocks, stream-K, ragged last tile block
    (5, 3, 5, 64, 64, False, False),       # tiny odd map, linear; a 1 x 5 strip mosaic: tiles straddle images
    (1, 2, 2, 96, 32, True, False),        # one tile per image; Cin not a power of two
    (2, 40, 36, 32, 64, True, True),       # the 208x208 stage's shape class: 4 K-steps per block
]


@pytest.mark.parametrize('n,h,w,cin,cout,act,resid', WINO_CASES)
def test_winograd_conv_matches_fp64(n, h, w, cin, cout, act, resid):
    """y3_conv2d_fwd_wino (F(2x2,3x3), exact fp32 arithmetic) against the fp64 reference at the direct kernel's
    tolerance, and against the direct kernel itself."""
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    rng = np.random.RandomState(n * 100 + h + cin)
    x, wt, scale, shift = make_case(rng, n, h, w, 3, cin, cout)
    r = rng.standard_normal((n, h, w, cout)).astype(np.float32) if resid else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, 3, 1, 1)
    assert _lib.lib().y3_conv_wino_eligible(d) == 1
    wu = engine.pack_wino(t(wt))
    got = engine.conv2d_fwd_wino(t(x), wu, t(scale), t(shift), cout, act, residual=t(r))
    torch.cuda.synchronize()
    want = ref_conv(x, wt, scale, shift, 3, 1, act, r)
    check(got.cpu().numpy(), want, 'winograd %dx%dx%d %d->%d' % (n, h, w, cin, cout))
    direct = run_gpu(x, wt, scale, shift, 3, 1, act, r)
    assert np.abs(got.cpu().numpy() - direct).max() <= 2e-4 * (1 + np.abs(want).max())
    # the one-workgroup-per-block schedule (no workspace) against the same reference, and run-to-run determinism
    plain = engine.conv2d_fwd_wino(t(x), wu, t(scale), t(shift), cout, act, residual=t(r), use_workspace=False)
    check(plain.cpu().numpy(), want, 'winograd (no workspace) %dx%dx%d %d->%d' % (n, h, w, cin, cout))
    again = engine.conv2d_fwd_wino(t(x), wu, t(scale), t(shift), cout, act, residual=t(r))
    assert torch.equal(again, got)


WINO44_CASES = [
    # n, h, w, cin, cout, act, resid
    (3, 20, 28, 64, 128, True, True),      # whole 4x4 tiles, ragged last tile block
    (2, 13, 13, 512, 1024, True, True),    # odd map: tiled as a mosaic of the two images (27 x 13 pixels, a zero gap row between them)
    (1, 26, 26, 256, 512, True, False),    # 7x7 tiles, the last half outside
    (2, 52, 52, 128, 256, True, True),     # true-size 52x52 residual-stage conv
    (5, 3, 5, 64, 64, False, False),       # tiny odd map, linear
    (1, 2, 2, 96, 64, True, False),        # one tile per image, mostly padding; Cin not a power of two (six 16-channel stages)
    (2, 40, 36, 32, 64, True, True),       # the 208x208 stage's shape class: 4 K-steps = two 16-channel stages per block
    (1, 9, 130, 32, 192, True, True),      # wide and flat, Cout = 3 column blocks
    (32, 26, 26, 64, 512, True, True),     # mosaic of 4 x 8 images: 1,458 tiles = 92 tile blocks x 8 channel blocks (92 = 11.5 per XCD: the XCD walk's ragged end)
    (10, 52, 52, 32, 128, True, False),    # 106 tile blocks x 2
    (10, 52, 52, 32, 640, False, True),    # 106 x 10 channel blocks
    (24, 26, 26, 128, 512, True, True),    # mosaic of 3 x 8 images: 1,080 tiles = 68 x 8 blocks
    (7, 13, 13, 160, 64, True, True),      # prime batch: a 1 x 7 strip mosaic; ten stages; one channel block
]


@pytest.mark.parametrize('n,h,w,cin,cout,act,resid', WINO44_CASES)
def test_winograd_f4x4_conv_matches_fp64(n, h, w, cin, cout, act, resid):
    """y3_conv2d_fwd_wino44 (F(4x4,3x3), fp32 arithmetic) against the fp64 reference at the direct kernel's tolerance, in both
    of its forms: two kernels (input transform written once into the workspace, then the batched GEMMs - what y3_net_forward
    runs) and one kernel (no workspace: the transform inside the K-loop).  Same arithmetic in the same order: the two must
    agree to the last bits the compiler's contraction of the transform expressions leaves open (1e-5 of the output scale)."""
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    rng = np.random.RandomState(n * 100 + h + cin)
    x, wt, scale, shift = make_case(rng, n, h, w, 3, cin, cout)
    r = rng.standard_normal((n, h, w, cout)).astype(np.float32) if resid else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    assert engine.wino44_eligible(3, 1, cin, cout)
    d = _lib.ConvDesc(n, h, w, cin, 0, cout, 3, 1, 1)
    tiles = _lib.lib().y3_conv_stats_blocks(d, 2)
    assert _lib.lib().y3_conv_wino44_workspace_bytes(d) == tiles * 16 * cin * 36 * 4      # V: 36 positions per tile and channel
    wu = engine.pack_wino44(t(wt))
    got = engine.conv2d_fwd_wino44(t(x), wu, t(scale), t(shift), cout, act, residual=t(r))
    torch.cuda.synchronize()
    want = ref_conv(x, wt, scale, shift, 3, 1, act, r)
    err = np.abs(got.cpu().numpy() - want)
    print('F(4x4,3x3) %dx%dx%d %d->%d: max err %.3e (max |ref| %.2f)' % (n, h, w, cin, cout, err.max(), np.abs(want).max()))
    check(got.cpu().numpy(), want, 'winograd F(4x4) %dx%dx%d %d->%d' % (n, h, w, cin, cout))
    # poison the workspace: nothing may be carried between calls, every byte the GEMM kernel reads is rewritten
    engine._conv_scratch(dev, 16).fill_(0xFF)
    again = engine.conv2d_fwd_wino44(t(x), wu, t(scale), t(shift), cout, act, residual=t(r))
    assert torch.equal(again, got)                    # run-to-run bit-exact
    plain = engine.conv2d_fwd_wino44(t(x), wu, t(scale), t(shift), cout, act, residual=t(r), use_workspace=False)
    check(plain.cpu().numpy(), want, 'winograd F(4x4), one kernel %dx%dx%d %d->%d' % (n, h, w, cin, cout))
    assert np.abs(plain.cpu().numpy() - got.cpu().numpy()).max() <= 1e-5 * (1 + np.abs(want).max())


def test_winograd_f4x4_eligibility_and_errors():
    from yolov3_tensorflow_amd import engine, framework as fw
    dev = fw.default_device()
    for args, ok in [((3, 1, 64, 64), True), ((3, 1, 32, 128), True), ((3, 2, 64, 64), False), ((1, 1, 64, 64), False),
                     ((3, 1, 48, 64), False), ((3, 1, 64, 32), False), ((3, 1, 64, 255), False)]:
        assert engine.wino44_eligible(*args) == ok, args
    x = torch.zeros((1, 8, 8, 64), device=dev)
    with pytest.raises(ValueError):
        engine.conv2d_fwd_wino44(x, torch.zeros(36 * 64 * 32, device=dev), torch.ones(32, device=dev),
                                 torch.zeros(32, device=dev), 32, True)


def test_winograd_eligibility_and_errors():
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    L = _lib.lib()
    for args, ok in (((1, 8, 8, 64, 0, 64, 3, 1, 1), 1), ((1, 8, 8, 64, 0, 64, 3, 2, 1), 0),
                     ((1, 8, 8, 64, 0, 64, 1, 1, 1), 0), ((1, 8, 8, 32, 0, 64, 3, 1, 1), 1), ((1, 8, 8, 48, 0, 64, 3, 1, 1), 0),
                     ((1, 8, 8, 64, 0, 255, 3, 1, 1), 0), ((1, 8, 8, 96, 32, 64, 3, 1, 1), 0)):
        assert L.y3_conv_wino_eligible(_lib.ConvDesc(*args)) == ok, args
    dev = fw.default_device()
    x = torch.zeros((1, 8, 8, 48), device=dev)
    with pytest.raises(ValueError):
        engine.conv2d_fwd_wino(x, torch.zeros(16 * 48 * 64, device=dev), torch.ones(64, device=dev),
                               torch.zeros(64, device=dev), 64, True)


def test_streamk_timeout_is_loud():
    """A stream-K hand-off that times out must never return a wrong tensor with rc 0 (include/yolo355.h, y3_ctx_check).
    y3_debug_streamk_fault(1) makes the producers skip raising their flag and shortens the consumers' poll: the launch ends,
    the kernel ORs a code into the context's error word, the next call on the context refuses to launch (Y3_EHIP ->
    Y3Error), y3_ctx_check reports and clears it, and the context then works again, bit-exactly."""
    from yolov3_tensorflow_amd import engine, framework as fw, _lib
    dev = fw.default_device()
    rng = np.random.RandomState(5)
    n, h, w, cin, cout = 8, 52, 52, 128, 256           # 338 direct tiles, F(2x2) Winograd blocks: both schedules stream-K
    x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).to(dev)
    wt = torch.from_numpy((rng.standard_normal((3, 3, cin, cout)) * 0.03).astype(np.float32)).to(dev)
    ones, zeros = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    wp = torch.empty(9 * cout * cin, device=dev)
    _lib.check(_lib.lib().y3_pack_conv_weights(fw.context(), fw.ptr(wt), 3, cin, cout, fw.ptr(wp)))
    wu = engine.pack_wino(wt)
    runs = {'direct': lambda: engine.conv2d_fwd(x, wp, ones, zeros, 3, 1, cout, True),
            'wino': lambda: engine.conv2d_fwd_wino(x, wu, ones, zeros, cout, True)}
    fw.check_context()
    try:
        for name, run in runs.items():
            good = run()
            fw.check_context()
            _lib.lib().y3_debug_streamk_fault(1)
            run()                                      # launches; its result is garbage and says so:
            torch.cuda.synchronize()                   # (the kernel has to have run for the word to be set)
            with pytest.raises(_lib.Y3Error, match='stream-K hand-off timed out'):
                run()                                  # ... the next call on the context refuses to launch
            with pytest.raises(_lib.Y3Error, match='stream-K hand-off timed out'):
                fw.check_context()                     # ... and the explicit check reports it (and clears it)
            _lib.lib().y3_debug_streamk_fault(0)
            fw.check_context()
            assert torch.equal(run(), good), name
            fw.check_context()
    finally:
        _lib.lib().y3_debug_streamk_fault(0)
        try:
            fw.check_context()                         # never leave the session's context poisoned for later tests
        except _lib.Y3Error:
            pass


@pytest.mark.parametrize('n,h,w', [(2, 64, 96), (1, 70, 38), (3, 32, 32), (2, 416, 416)])
def test_fused_stem_and_stride2_conv(n, h, w):
    """The stem and the stride-2 conv behind it in one kernel (y3_conv2d_fwd_stem_s2, csrc/y3_conv_f32s.hip; what y3_net_forward
    runs for layers 0 and 1 in the fp32 modes) against the fp64 reference of the two convs (utils/layer_utils.py:34-40), at the
    tolerance of the separate kernels.  Maps that are no multiple of the 8 x 16 tile, several images, the bench's own size."""
    from yolov3_tensorflow_amd import framework as fw, _lib
    dev = fw.default_device()
    L, ctx = _lib.lib(), fw.context()
    rng = np.random.RandomState(h * 7 + w)
    x = rng.uniform(0, 1, (n, h, w, 3)).astype(np.float32)
    _, w0, sc0, sh0 = make_case(rng, 1, 1, 1, 3, 3, 32)
    _, w1, sc1, sh1 = make_case(rng, 1, 1, 1, 3, 32, 64)
    mid = ref_conv(x, w0, sc0, sh0, 3, 1, True)
    want = ref_conv(mid.astype(np.float64), w1, sc1, sh1, 3, 2, True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    w1g = t(w1)
    w1p = torch.empty(9 * 64 * 32, device=dev)
    _lib.check(L.y3_pack_conv_weights(ctx, fw.ptr(w1g), 3, 32, 64, fw.ptr(w1p)))
    y = torch.empty((n, h // 2, w // 2, 64), device=dev)
    xg, w0g, a0, b0, a1, b1 = t(x), t(w0), t(sc0), t(sh0), t(sc1), t(sh1)
    _lib.check(L.y3_conv2d_fwd_stem_s2(ctx, n, h, w, fw.ptr(xg), fw.ptr(w0g), fw.ptr(a0), fw.ptr(b0), fw.ptr(w1p), fw.ptr(a1),
                                       fw.ptr(b1), fw.ptr(y)))
    torch.cuda.synchronize()
    check(y.cpu().numpy(), want, 'fused stem + stride-2 conv %dx%dx%d' % (n, h, w))
