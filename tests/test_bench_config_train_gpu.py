"""Parity of the TRAIN step at the BASELINE configs[3] size (ref: train.py:105-115 at 416x416, bs=64 per GPU, the default
'f32_wino' mode).  The whole-step test of tests/test_train_gpu.py runs at 256 px / bs=4; the stream-K split points of the
forward and data-gradient convs, the Winograd dgrad / wgrad tilings and the block counts of the statistics epilogue all
depend on the batch and the map sizes, so the bench configuration gets its own comparisons (VERDICT r2, next #1a):

  * bs=64 @416: loss 5-tuple of the Winograd-mode training forward vs the fp64 CPU oracle (forward + loss only), 1e-4
    relative; the direct-kernel training forward ('f32') on the same batch: feature maps 2e-4 + 1e-4*|ref|, loss 1e-4;
  * bs=64 @416: EVERY gradient tensor of the Winograd-mode backward vs the direct-kernel backward, both run from the SAME
    saved forward state (y3_net_train_backward re-run with the net in another mode: same z, same BN statistics, hence the
    same LeakyReLU branches: the comparison is between kernels,
    not between branch patterns — see tests/test_train_gpu.py for why that matters), 2e-4 of the tensor's max magnitude;
    run-to-run bit-exactness of the Winograd-mode backward;
  * bs=8 @416 (the 13/26/52-grid map sizes of the bench): one whole step vs the fp64 autograd oracle evaluated on the
    LeakyReLU branches the GPU took — loss 1e-4, every gradient tensor 2e-4 (the tolerances of the 256 px test).
"""
import numpy as np
import pytest
import torch

from conftest import COCO_ANCHORS, blob_images

pytestmark = pytest.mark.gpu

SIZE = 416
GRAD_TOL = 2e-4


def rel_err(got, want):
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-12))


def _fresh_model(params, **kw):
    import yolov3_tensorflow_amd as y3
    y3.reset_default_graph()
    model = y3.yolov3(80, COCO_ANCHORS, **kw)
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 32, 32, 3))
    for v in y3.global_variables(scope='yolov3'):
        v.assign(params[v.op_name])
    return model


def _conv_name(i):
    sub, j = ('darknet53_body', i) if i < 52 else ('yolov3_head', i - 52)
    return 'yolov3/%s/%s' % (sub, 'Conv' if j == 0 else 'Conv_%d' % j)


def test_configs3_bs64_416_loss_and_every_gradient(isolated_graph):
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from oracle import yolo_ref, train_ref
    n = 64
    params = yolo_ref.synthetic_params(80, seed=1)
    x = blob_images(31, n, SIZE)
    yts = train_ref.synthetic_targets(7, n, [SIZE, SIZE], 80, COCO_ANCHORS, max_boxes=6)
    model = _fresh_model(params, batch_norm_decay=0.99, weight_decay=5e-4)
    xd = torch.from_numpy(x).cuda()
    ytd = [torch.from_numpy(np.asarray(y, np.float32)).cuda() for y in yts]

    # ---- the fp64 oracle: forward (batch-statistics BN) + loss, no autograd ----
    torch.set_num_threads(min(torch.get_num_threads(), 64))
    g = train_ref.TrainGraph(params, dtype=torch.float64)
    g.keep_trace = False
    with torch.no_grad():
        ref_fms = g.forward(x)                                    # NHWC
        ref_loss = [float(v) for v in g.compute_loss(ref_fms, yts, COCO_ANCHORS)]
    ref_fms = [f.contiguous().numpy() for f in ref_fms]
    del g

    # ---- Winograd mode (the bench's): forward + loss, then backward twice from the same saved state ----
    model.compute_dtype = 'f32_wino'
    trainer = training.Trainer(model, config_optimizer('sgd', 1e-4))
    with y3.variable_scope('yolov3'):
        fms_w = [f.clone() for f in model.forward(xd, is_training=True)]       # (views of the step's workspace: keep copies)
        loss_w = [float(v) for v in training.compute_loss(model, model._train['fms'], ytd)]

        def backward_with(dtype):
            # y3_net_train_backward is re-runnable from the forward / loss state; the net's mode only selects the kernels
            model.compute_dtype = dtype
            trainer.backward()
            torch.cuda.synchronize()
            return trainer.flat.clone()

        gw = backward_with('f32_wino')
        gw2 = backward_with('f32_wino')
        gd = backward_with('f32')
    for a, b in zip(loss_w, ref_loss):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, ('f32_wino loss vs fp64 oracle', loss_w, ref_loss)
    worst_fm = 0.0
    for i, (a, r) in enumerate(zip(fms_w, ref_fms)):
        err = np.abs(a.cpu().numpy() - r)
        assert (err <= 2e-4 + 1e-4 * np.abs(r)).all(), 'train-mode feature map %d vs fp64 oracle: %.3e' % (i + 1, err.max())
        worst_fm = max(worst_fm, float(err.max()))
    assert torch.equal(gw, gw2), 'bs=64 Winograd-mode backward is not run-to-run bit-exact'
    errs = {}
    for v in trainer.order:
        o = trainer.offsets[v.op_name]
        k = v.tensor.numel()
        errs[v.op_name] = rel_err(gw[o:o + k].cpu().numpy(), gd[o:o + k].cpu().numpy())
    worst = max(errs, key=errs.get)
    print('bs=64 @416: loss %s vs fp64 oracle %s; train-mode feature maps max |d| %.2e; Winograd-mode vs direct-kernel '
          'gradients from the same forward state: worst %.2e (%s), median %.2e over %d tensors'
          % (['%.6g' % v for v in loss_w], ['%.6g' % v for v in ref_loss], worst_fm, errs[worst], worst,
             float(np.median(list(errs.values()))), len(errs)))
    assert len(errs) == 222
    for name, e in errs.items():
        assert e < GRAD_TOL, '%s: Winograd-mode vs direct-kernel gradient rel err %.3e' % (name, e)

    # ---- direct-kernel training forward on the same batch ----
    model.compute_dtype = 'f32'
    with y3.variable_scope('yolov3'):
        fms_d = model.forward(xd, is_training=True)
        loss_d = [float(v) for v in training.compute_loss(model, fms_d, ytd)]
    for a, b in zip(loss_d, ref_loss):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, ('f32 loss vs fp64 oracle', loss_d, ref_loss)
    for i, (a, r) in enumerate(zip(fms_d, ref_fms)):
        err = np.abs(a.cpu().numpy() - r)
        assert (err <= 2e-4 + 1e-4 * np.abs(r)).all(), 'direct train-mode feature map %d: %.3e' % (i + 1, err.max())


def test_configs3_bs64_416_gradients_vs_fp64_autograd(isolated_graph):
    """the HIP-vs-HIP comparison above would pass a bug COMMON to both backward families at the 64-image
    split points (BN-backward reductions, the weight-gradient split sums, the loss gradient).  Here the whole bs=64 @416
    step is differentiated by fp64 autograd on the host (the oracle, on the LeakyReLU branches the GPU took - see
    tests/test_train_gpu.py for why) and EVERY gradient tensor of the 'f32_wino' step is held against it: 2e-4 of the
    tensor's max magnitude, the tolerance of the 256 px / bs=4 and 416 px / bs=8 tests.  Slow (the fp64 backward of 64
    images runs on the host cores): several minutes, ~100 GB of host memory."""
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from oracle import yolo_ref, train_ref
    import os
    try:
        avail_gb = os.sysconf('SC_AVPHYS_PAGES') * os.sysconf('SC_PAGE_SIZE') / 2.0 ** 30
    except (ValueError, OSError):
        avail_gb = 0.0
    if avail_gb < 160 or (os.cpu_count() or 1) < 32:
        pytest.skip('the fp64 autograd oracle at bs=64 @416 needs ~100 GB of host memory and many cores (%.0f GB, %d cores here)'
                    % (avail_gb, os.cpu_count() or 1))
    n, lr = 64, 1e-4
    params = yolo_ref.synthetic_params(80, seed=1)
    x = blob_images(31, n, SIZE)
    yts = train_ref.synthetic_targets(7, n, [SIZE, SIZE], 80, COCO_ANCHORS, max_boxes=6)
    model = _fresh_model(params, batch_norm_decay=0.99, weight_decay=5e-4)
    model.compute_dtype = 'f32_wino'
    trainer = training.Trainer(model, config_optimizer('sgd', lr))
    trainer.capture = []
    with y3.variable_scope('yolov3'):
        loss = trainer.step(x, yts)
    masks = {}
    for rec in trainer.capture:
        if rec['z'] is None:
            continue
        pos = (rec['z'] * rec['stats'][2] + rec['stats'][3]) > 0
        masks[_conv_name(rec['layer'])] = pos.permute(0, 3, 1, 2).cpu()
    trainer.capture = None
    grads = {k: v.cpu().numpy() for k, v in trainer.views.items()}
    loss = [float(v) for v in loss]
    del trainer
    torch.cuda.empty_cache()
    torch.set_num_threads(min(os.cpu_count() or 1, 128))
    ref = train_ref.train_step(params, x, yts, COCO_ANCHORS, optimizer='sgd', lr=lr, weight_decay=5e-4, bn_decay=0.99,
                               dtype=torch.float64, step=1, masks=masks)
    for a, b in zip(loss, ref['loss']):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (loss, ref['loss'])
    assert set(grads) == set(ref['grads']) and len(grads) == 222
    errs = {name: rel_err(grads[name], gr) for name, gr in ref['grads'].items()}
    worst = max(errs, key=errs.get)
    head = [e for k, e in errs.items() if '/yolov3_head/' in k]
    stem = [errs['yolov3/darknet53_body/%s/weights' % c] for c in ('Conv', 'Conv_1', 'Conv_2')]
    bn = [e for k, e in errs.items() if k.endswith('/gamma') or k.endswith('/beta')]
    print('bs=64 @416 f32_wino vs fp64 autograd on the GPU\'s branches: worst %.2e (%s), median %.2e over %d tensors; head '
          'worst %.2e, first three body convs %s, BN gamma/beta worst %.2e'
          % (errs[worst], worst, float(np.median(list(errs.values()))), len(errs), max(head),
             ['%.2e' % e for e in stem], max(bn)))
    for name, e in errs.items():
        assert e < GRAD_TOL, '%s: grad rel err %.3e' % (name, e)


def test_one_train_step_at_416_bs8_matches_oracle(isolated_graph):
    """The whole-step comparison of tests/test_train_gpu.py at the bench's map sizes (416 px: 13/26/52/104/208 grids)."""
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    from oracle import yolo_ref, train_ref
    n, lr = 8, 1e-3
    params = yolo_ref.synthetic_params(80, seed=1)
    x = blob_images(33, n, SIZE)
    yts = train_ref.synthetic_targets(9, n, [SIZE, SIZE], 80, COCO_ANCHORS, max_boxes=5)
    model = _fresh_model(params, batch_norm_decay=0.99, weight_decay=5e-4)
    model.compute_dtype = 'f32_wino'
    trainer = training.Trainer(model, config_optimizer('sgd', lr))
    trainer.capture = []
    with y3.variable_scope('yolov3'):
        loss = trainer.step(x, yts)
    masks = {}
    for rec in trainer.capture:
        if rec['z'] is None:
            continue
        pos = (rec['z'] * rec['stats'][2] + rec['stats'][3]) > 0
        masks[_conv_name(rec['layer'])] = pos.permute(0, 3, 1, 2).cpu()
    trainer.capture = None
    grads = {k: v.cpu().numpy() for k, v in trainer.views.items()}
    torch.set_num_threads(min(torch.get_num_threads(), 64))
    ref = train_ref.train_step(params, x, yts, COCO_ANCHORS, optimizer='sgd', lr=lr, weight_decay=5e-4, bn_decay=0.99,
                               dtype=torch.float64, step=1, masks=masks)
    for a, b in zip(loss, ref['loss']):
        assert abs(float(a) - b) <= 1e-4 * abs(b) + 1e-6, ([float(v) for v in loss], ref['loss'])
    assert set(grads) == set(ref['grads'])
    errs = {name: rel_err(grads[name], gr) for name, gr in ref['grads'].items()}
    worst = max(errs, key=errs.get)
    print('bs=8 @416 f32_wino: gradient rel err vs fp64 oracle on the GPU\'s branches: worst %.2e (%s), median %.2e over %d '
          'tensors' % (errs[worst], worst, float(np.median(list(errs.values()))), len(errs)))
    for name, e in errs.items():
        assert e < GRAD_TOL, '%s: grad rel err %.3e' % (name, e)
    for v in y3.global_variables(scope='yolov3'):
        want = ref['new_params'][v.op_name]
        scale = max(np.abs(want).max(), 1e-6)
        slack = GRAD_TOL * lr * float(np.abs(ref['grads'][v.op_name]).max()) if v.op_name in ref['grads'] else 0.0
        assert np.abs(v.numpy() - want).max() <= 1e-4 * scale + slack, v.op_name
