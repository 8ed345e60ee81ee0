"""CPU, world_size-2 `gloo` tests of the N>1 logic: the data-parallel gradient exchange is ONE
all-reduce of the flat gradient buffer followed by a 1/world scale (folded into the clip/update kernel on
the GPU), and inference replicas need no exchange at all — each rank derives its own shard of the images."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, init_file, out_dir):
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    # each rank's "local gradients": deterministic, rank dependent
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    flat = g.clone()
    dist.all_reduce(flat)                       # what Trainer.apply_gradients does with its flat buffer
    mean = flat * (1.0 / world)                 # the grad_scale handed to y3_clip_update
    # a per-tensor clip after averaging must see the same norm on every rank
    norm = float(torch.sqrt((mean[:100] ** 2).sum()))
    # inference replicas: rank r takes images r, r+world, ... (no collective); the union must be the batch
    images = list(range(rank, 10, world))
    t = torch.tensor([float(rank == 0) * 3.0 + 1.0])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)    # bench.py: max over ranks of the elapsed time
    torch.save(dict(mean=mean, norm=norm, images=images, tmax=float(t)), os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_and_sharding():
    d = tempfile.mkdtemp()
    init = os.path.join(d, 'init')
    mp.spawn(_worker, args=(2, init, d), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(d, 'r0.pt')), torch.load(os.path.join(d, 'r1.pt'))
    want = torch.arange(1000, dtype=torch.float32) * 1.5          # mean of g*1 and g*2
    assert torch.equal(r0['mean'], want) and torch.equal(r1['mean'], want)
    assert r0['norm'] == r1['norm']
    assert sorted(r0['images'] + r1['images']) == list(range(10)) and not set(r0['images']) & set(r1['images'])
    assert r0['tmax'] == r1['tmax'] == 4.0


def test_trainer_is_single_process_without_a_process_group():
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    opt = config_optimizer('momentum', 1e-3)
    assert isinstance(opt, training.Optimizer) and opt.kind == 'momentum' and opt.momentum == 0.9
    assert config_optimizer('rmsprop', 1e-3).epsilon == 1e-10 and config_optimizer('adam', 1e-3).epsilon == 1e-8
    assert opt.lr_at(5.0) == 1e-3
    assert training.Optimizer('sgd', lambda step: 0.1 * step).lr_at(3.0) == pytest.approx(0.3)
    assert not dist.is_initialized()
