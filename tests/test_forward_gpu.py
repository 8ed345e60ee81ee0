"""GPU parity of yolov3.forward (y3_net_forward through the C ABI) against the CPU oracle, with the
synthetic weights loaded through the darknet-format loader (BASELINE config 3 wiring).
Tolerances (stated): feature maps |d| <= 2e-4 + 1e-4*|ref| against the fp64 oracle (the fp32 oracle itself
sits ~5e-6 from fp64); fused plan vs op-by-op composition: bit-exact.
The oracle comparisons run for compute_dtype 'f32' (exact fp32 MFMA) and 'f32_bf16x6' (fp32 products rebuilt from
six bf16 plane products) and 'f32_wino' (Winograd F(2x2,3x3) for the stride-1 3x3 convs) at the SAME tolerances."""
import numpy as np
import pytest
import torch

from conftest import blob_images

pytestmark = pytest.mark.gpu


def _cmp(got, want, what, atol=2e-4, rtol=1e-4):
    err = np.abs(got - want)
    assert got.shape == want.shape
    assert (err <= atol + rtol * np.abs(want)).all(), '%s: max err %.3e' % (what, err.max())
    return float(err.max())


DTYPES = ['f32', 'f32_bf16x6', 'f32_wino']


@pytest.fixture
def with_dtype(gpu_model):
    model, _ = gpu_model

    def setter(dtype):
        model.compute_dtype = dtype
    yield setter
    model.compute_dtype = 'f32'


@pytest.mark.parametrize('dtype', DTYPES)
def test_forward_matches_oracle_416(gpu_model, with_dtype, dtype):
    import yolov3_tensorflow_amd as y3
    from oracle import yolo_ref
    model, params = gpu_model
    with_dtype(dtype)
    x = blob_images(0, 2, 416)
    with y3.variable_scope('yolov3'):
        fms = model.forward(x, False)
    torch.cuda.synchronize()
    ref64 = yolo_ref.forward(params, x, dtype=torch.float64)
    ref32 = yolo_ref.forward(params, x, dtype=torch.float32)
    for i, (g, r64, r32) in enumerate(zip(fms, ref64, ref32)):
        g = g.cpu().numpy()
        e = _cmp(g, r64, 'feature_map_%d vs fp64 oracle' % (i + 1))
        e32 = float(np.abs(r32 - r64).max())
        print('feature_map_%d: GPU-vs-fp64 %.3e ; oracle-fp32-vs-fp64 %.3e' % (i + 1, e, e32))
        assert e <= max(20 * e32, 5e-5), 'GPU drift is far above the fp32 CPU drift'


@pytest.mark.parametrize('dtype', DTYPES)
def test_forward_other_sizes_and_batches(gpu_model, with_dtype, dtype):
    import yolov3_tensorflow_amd as y3
    from oracle import yolo_ref
    model, params = gpu_model
    with_dtype(dtype)
    for n, h, w in ((1, 320, 608), (3, 96, 64)):
        rng = np.random.RandomState(h)
        x = rng.rand(n, h, w, 3).astype(np.float32)
        with y3.variable_scope('yolov3'):
            fms = model.forward(x, False)
        ref = yolo_ref.forward(params, x, dtype=torch.float64)
        assert model.img_size == [h, w]
        for g, r in zip(fms, ref):
            _cmp(g.cpu().numpy(), r, 'forward %dx%dx%d' % (n, h, w))


def test_fused_plan_equals_op_by_op_composition(gpu_model):
    import yolov3_tensorflow_amd as y3
    model, _ = gpu_model
    x = torch.from_numpy(blob_images(3, 2, 224)).cuda()
    with y3.variable_scope('yolov3'):
        fused = model.forward(x, False)
        composed = model.forward_composed(x)
    for a, b in zip(fused, composed):
        assert torch.equal(a, b)
    # and no new variables were created by the composed path (same names, same order)
    assert len(y3.global_variables(scope='yolov3')) == 366


def test_fused_launches_are_reported_and_cost_no_layer_time(gpu_model, with_dtype):
    """y3_net_layer_fused (include/yolo355.h): which launches the plan fuses at a size - the stem with the stride-2 conv behind it
    (utils/layer_utils.py:34-40) in the fp32 and bf16 paths, the first residual block in the bf16 path - and the profiled time of
    a layer that runs inside the next layer's launch is the time between two adjacent events: nothing.  bench.py's algorithmic
    bytes follow this report (a fused pair is ONE kernel's bytes: the tensor between them never reaches memory)."""
    import yolov3_tensorflow_amd as y3
    import bench
    model, _ = gpu_model
    x = torch.from_numpy(blob_images(2, 2, 416)).cuda()
    for dtype, want in (('f32_wino', {0: 1, 1: 2}), ('bf16', {0: 1, 1: 2, 2: 1, 3: 2}), ('f32', {0: 1, 1: 2})):
        with_dtype(dtype)
        with y3.variable_scope('yolov3'):
            model.forward(x, False)
            fused = model.layer_fused(2, 416, 416)
            assert {i: int(v) for i, v in enumerate(fused) if v} == want, (dtype, fused)
            ms, table = model.layer_times_ms(x, iters=3)
        for i, v in enumerate(fused):
            if v == 1:
                assert ms[i] < 0.02, (dtype, i, ms[i], ms[i + 1])      # (two adjacent events on the stream: a few us, no kernel)
        plain = bench.conv_bytes(table, 2, 416, 416, 2 if dtype == 'bf16' else 4)
        aware = bench.conv_bytes(table, 2, 416, 416, 2 if dtype == 'bf16' else 4, fused)
        es = 2 if dtype == 'bf16' else 4
        stem_out = 2 * 416 * 416 * 32 * es
        assert plain[0] - aware[0] == stem_out and plain[1] - aware[1] == stem_out          # written once, read once: neither happens
        if dtype == 'bf16':
            mid = 2 * 208 * 208 * 32 * es
            assert plain[2] - aware[2] == mid and plain[3] - aware[3] == mid + 2 * 208 * 208 * 64 * es      # ... and the shortcut is the block's input
        assert (plain[4:] == aware[4:]).all()


def test_variables_follow_reference_naming_and_order(gpu_model):
    import yolov3_tensorflow_amd as y3
    from oracle import yolo_ref
    names = [v.op_name for v in y3.global_variables(scope='yolov3')]
    assert names == [n for n, _ in yolo_ref.variable_specs(80)]
    shapes = [tuple(v.shape) for v in y3.global_variables(scope='yolov3')]
    assert shapes == [tuple(s) for _, s in yolo_ref.variable_specs(80)]


def test_forward_is_deterministic_and_batch_independent(gpu_model):
    import yolov3_tensorflow_amd as y3
    model, _ = gpu_model
    x = torch.from_numpy(blob_images(5, 4, 256)).cuda()
    with y3.variable_scope('yolov3'):
        a = [t.clone() for t in model.forward(x, False)]
        b = model.forward(x, False)
        single = model.forward(x[2:3].contiguous(), False)
    for p, q, s in zip(a, b, single):
        assert torch.equal(p, q)                               # run-to-run: bit-exact
        # batch independence holds to fp32 rounding (stream-K split points depend on the batch size)
        assert torch.allclose(p[2], s[0], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dtype', ['f32', 'f32_wino'])
def test_forward_on_two_streams_equals_one_stream(gpu_model, with_dtype, dtype):
    """model.inference_streams = 2 (BASELINE north star: "independent per-GPU streams for inference"): the batch as two
    halves on two HIP streams, each through its own net / context / workspace, writing slices of the same outputs.  Same
    kernels on the same images: equal to the one-stream forward to fp32 rounding (stream-K split points and the kernel
    choice depend on the batch size), run-to-run bit-exact, no cross-stream hazard over repeated calls, and a forward
    with layer profiling on stays on one stream."""
    import yolov3_tensorflow_amd as y3
    model, _ = gpu_model
    with_dtype(dtype)
    x = torch.from_numpy(blob_images(9, 8, 256)).cuda()
    try:
        with y3.variable_scope('yolov3'):
            one = [t.clone() for t in model.forward(x, False)]
            model.inference_streams = 2
            runs = [[t.clone() for t in model.forward(x, False)] for _ in range(3)]
            torch.cuda.synchronize()
            from yolov3_tensorflow_amd import model as model_module
            assert len(model_module._SIDE_STREAMS[(x.device.index, 2)]) == 1        # one set per process, shared by every model
            model.set_layer_profiling(True)
            prof = [t.clone() for t in model.forward(x, False)]
            model.set_layer_profiling(False)
            model.read_layer_ms()
        for a, b, c, p, o in zip(runs[0], runs[1], runs[2], prof, one):
            assert torch.equal(a, b) and torch.equal(a, c)
            assert torch.allclose(a, o, rtol=1e-4, atol=1e-4)
            assert torch.equal(p, o)                      # profiled forward = the one-stream path
    finally:
        model.inference_streams = 1


def test_bad_input_raises(gpu_model):
    import yolov3_tensorflow_amd as y3
    model, _ = gpu_model
    with y3.variable_scope('yolov3'):
        with pytest.raises(ValueError):
            model.forward(np.zeros((1, 100, 100, 3), np.float32))    # not a multiple of 32
        with pytest.raises(ValueError):
            model.forward(np.zeros((1, 64, 64, 4), np.float32))      # not RGB


def test_rebuilt_graph_never_sees_the_previous_graphs_packed_weights(isolated_graph):
    """Regression: the packed-weight caches are keyed by (variable name, version).  Two networks built one after the
    other under the same names (reset_default_graph in between) with different weights must each produce their own
    outputs — versions are unique across the process, not per variable."""
    import yolov3_tensorflow_amd as y3
    from conftest import COCO_ANCHORS
    from oracle import yolo_ref
    x = blob_images(5, 1, 96)
    outs, seen = [], set()
    for seed in (3, 4):
        y3.reset_default_graph()
        params = yolo_ref.synthetic_params(80, seed=seed)
        model = y3.yolov3(80, COCO_ANCHORS)
        with y3.variable_scope('yolov3'):
            model.forward(torch.zeros(1, 32, 32, 3))
            for v in y3.global_variables(scope='yolov3'):
                v.assign(params[v.op_name])
                assert v.version not in seen
                seen.add(v.version)
            fms = model.forward(x, False)
        ref = yolo_ref.forward(params, x, dtype=torch.float64)
        for g, r in zip(fms, ref):
            _cmp(g.cpu().numpy(), r, 'rebuilt graph, seed %d' % seed)
        outs.append(fms[0].cpu().numpy())
    assert np.abs(outs[0] - outs[1]).max() > 1e-3


def test_inference_after_head_only_training_uses_the_updated_moving_statistics(isolated_graph):
    """Regression: a training-mode forward updates the moving mean/variance of EVERY BN layer in place, also of layers
    whose gamma/beta are frozen (update_part = head): the folded inference parameters must follow."""
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from conftest import COCO_ANCHORS
    from oracle import yolo_ref, train_ref
    params = yolo_ref.synthetic_params(80, seed=6)
    model = y3.yolov3(80, COCO_ANCHORS, batch_norm_decay=0.5)
    x = blob_images(8, 2, 96)
    yts = train_ref.synthetic_targets(9, 2, [96, 96], 80, COCO_ANCHORS, max_boxes=3)
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 32, 32, 3))
        for v in y3.global_variables(scope='yolov3'):
            v.assign(params[v.op_name])
        before = [f.clone() for f in model.forward(x, False)]
        head = [v for v in y3.global_variables(scope='yolov3') if v.op_name.startswith('yolov3/yolov3_head')]
        trainer = training.Trainer(model, config_optimizer('sgd', 0.0), update_vars=head)     # lr 0: only BN stats move
        trainer.step(x, yts)
        after = model.forward(x, False)
    new_params = {v.op_name: v.numpy() for v in y3.global_variables(scope='yolov3')}
    assert np.abs(new_params['yolov3/darknet53_body/Conv/BatchNorm/moving_mean'] -
                  params['yolov3/darknet53_body/Conv/BatchNorm/moving_mean']).max() > 1e-4
    ref = yolo_ref.forward(new_params, x, dtype=torch.float64)
    for g, r, b in zip(after, ref, before):
        _cmp(g.cpu().numpy(), r, 'inference after head-only training step')
        assert float((g - b).abs().max()) > 1e-4


def test_choose_inference_streams_keeps_a_measured_candidate(gpu_model):
    """yolov3.choose_inference_streams: times the forward with each candidate stream count and leaves the fastest in
    model.inference_streams (how bench.py guards the two-stream mode against a runtime that maps both streams to one
    hardware queue)."""
    import yolov3_tensorflow_amd as y3
    model, _ = gpu_model
    x = torch.from_numpy(blob_images(3, 8, 128)).cuda()
    try:
        with y3.variable_scope('yolov3'):
            best = model.choose_inference_streams(x, candidates=(1, 2), iters=2)
            assert best in (1, 2) and model.inference_streams == best
            out = model.forward(x, False)
        assert all(torch.isfinite(t).all().item() for t in out)
    finally:
        model.inference_streams = 1
