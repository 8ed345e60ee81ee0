"""The data-parallel train step on DEVICE tensors with two ranks (north_star: "RCCL all-reduce of gradients").
With two or more GPUs visible each rank takes its own device and the process group is `nccl` (= RCCL over xGMI: the
production transport); on the one-MI355X boxes this suite usually runs on both ranks share cuda:0 and the group is `gloo`
(RCCL refuses two ranks on one device) — the product code path is the same either way: `training.Trainer(process_group=...)` ->
`distributed.GradientExchange` issuing bucketed asynchronous all-reduces on slices of the flat device gradient buffer
while backward is still launching kernels, `grad_scale = 1/world` folded into `y3_clip_update_multi`.

Checked: both ranks end with bit-identical variables; they equal what a single process computes by hand from the two
ranks' local gradients (mean, then clip, then update); several buckets were in flight before backward finished."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import COCO_ANCHORS, blob_images

pytestmark = pytest.mark.gpu

N, SIZE, LR = 2, 96, 1e-3


def _setup(seed_params=3):
    import yolov3_tensorflow_amd as y3
    from oracle import yolo_ref
    params = yolo_ref.synthetic_params(80, seed=seed_params)
    y3.reset_default_graph()
    model = y3.yolov3(80, COCO_ANCHORS, batch_norm_decay=0.9, weight_decay=5e-4)
    model.compute_dtype = 'f32_wino'
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 32, 32, 3))
    for v in y3.global_variables(scope='yolov3'):
        v.assign(params[v.op_name])
    return y3, model


def _data(rank):
    from oracle import train_ref
    x = blob_images(40 + rank, N, SIZE)
    yts = train_ref.synthetic_targets(50 + rank, N, [SIZE, SIZE], 80, COCO_ANCHORS, max_boxes=3)
    return x, yts


def _backend_and_device(rank):
    """nccl (RCCL) with one device per rank when the box has at least two GPUs, gloo on the shared cuda:0 otherwise."""
    if torch.cuda.device_count() >= 2:
        return 'nccl', rank
    return 'gloo', 0


def _worker(rank, world, init_file, out_dir):
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    import yolov3_tensorflow_amd as y3_
    backend, device = _backend_and_device(rank)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(device)
    y3_.set_default_device('cuda:%d' % device)
    dist.init_process_group(backend, init_method='file://' + init_file, rank=rank, world_size=world)
    y3, model = _setup()
    trainer = training.Trainer(model, config_optimizer('momentum', LR), process_group=dist.group.WORLD,
                               bucket_bytes=16 << 20)
    x, yts = _data(rank)
    issued_before_end = []
    with y3.variable_scope('yolov3'):
        fms = model.forward(x, is_training=True)
        training.compute_loss(model, fms, yts)
        trainer.backward()
        issued_before_end.append(len(trainer.exchange.issued))      # buckets issued while backward was running
        trainer.apply_gradients()
    torch.cuda.synchronize()
    state = {v.op_name: v.numpy() for v in y3.global_variables(scope='yolov3')}
    np.savez(os.path.join(out_dir, 'r%d.npz' % rank), **state)
    torch.save(dict(issued_before_end=issued_before_end, buckets=len(trainer.exchange.edges), world=trainer.exchange.world,
                    backend=backend),
               os.path.join(out_dir, 'm%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def _replay(_, out_dir):
    """Local gradients of the two shards by hand -> mean -> clip -> momentum update, in one process."""
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    torch.cuda.set_device(0)
    grads = []
    for rank in range(2):
        y3, model = _setup()
        trainer = training.Trainer(model, config_optimizer('momentum', LR))
        x, yts = _data(rank)
        with y3.variable_scope('yolov3'):
            fms = model.forward(x, is_training=True)
            training.compute_loss(model, fms, yts)
            trainer.backward()
        grads.append(trainer.flat.clone())
    y3, model = _setup()
    trainer = training.Trainer(model, config_optimizer('momentum', LR))
    x, yts = _data(0)
    with y3.variable_scope('yolov3'):
        fms = model.forward(x, is_training=True)
        training.compute_loss(model, fms, yts)
        trainer.backward()
        trainer.flat.copy_((grads[0] + grads[1]) * 0.5)       # what the exchange + grad_scale produce
        trainer.apply_gradients()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, 'replay.npz'), **{v.op_name: v.numpy() for v in y3.global_variables(scope='yolov3')})


def test_two_rank_train_steps_on_device_tensors():
    d = tempfile.mkdtemp()
    mp.spawn(_worker, args=(2, os.path.join(d, 'init'), d), nprocs=2, join=True)
    r0, r1 = np.load(os.path.join(d, 'r0.npz')), np.load(os.path.join(d, 'r1.npz'))
    m0 = torch.load(os.path.join(d, 'm0.pt'))
    assert m0['world'] == 2 and m0['buckets'] >= 8
    assert m0['backend'] == ('nccl' if torch.cuda.device_count() >= 2 else 'gloo')
    print('two-rank step ran on backend %s' % m0['backend'])
    assert all(k >= 2 for k in m0['issued_before_end']), 'no bucket was issued before backward finished: nothing overlaps'
    # the trainable variables end identical on both ranks (the BN moving statistics are per rank: no sync-BN, like the reference)
    for k in r0.files:
        if k.endswith(('moving_mean', 'moving_variance')):
            continue
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    assert np.abs(r0['yolov3/darknet53_body/Conv/BatchNorm/moving_mean'] -
                  r1['yolov3/darknet53_body/Conv/BatchNorm/moving_mean']).max() > 0       # different shards, different statistics

    # single-process replay of the step (in its own process: it resets the default graph, which the session-scoped
    # model of the other test files lives in)
    mp.spawn(_replay, args=(d,), nprocs=1, join=True)
    after1 = np.load(os.path.join(d, 'replay.npz'))
    for k in after1.files:
        if k.endswith(('moving_mean', 'moving_variance')):
            continue
        np.testing.assert_allclose(r0[k], after1[k], rtol=2e-6, atol=1e-7, err_msg=k)
