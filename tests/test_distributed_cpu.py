"""CPU, world_size-2 `gloo` tests of the N>1 logic driving the PRODUCT's gradient-exchange code:
`training.gradient_layout` (flat gradient buffer in the order backward produces the gradients) and
`distributed.GradientExchange` (bucketed all-reduce issued while backward is still running, 1/world folded into the
consumer) — the same objects `training.Trainer` uses on RCCL, here on CPU tensors — plus bench.py's multi-rank launch
logic (`--gpus N` re-executes itself under torch.distributed.run)."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Var(object):
    """Shape-only stand-in for framework.Variable (a real one needs a HIP device)."""

    def __init__(self, op_name, shape, trainable=True):
        self.op_name, self.shape, self.trainable = op_name, tuple(shape), trainable


def _layer_vars(class_num=80):
    """The 75 layers' variables with the product's names and shapes, from the library's own layer table."""
    from yolov3_tensorflow_amd import training
    topo = training._Topology(class_num)           # host-only: no device needed
    out = []
    for i, l in enumerate(topo.layers):
        sub, j = ('darknet53_body', i) if i < 52 else ('yolov3_head', i - 52)
        base = 'yolov3/%s/%s' % (sub, 'Conv' if j == 0 else 'Conv_%d' % j)
        w = _Var(base + '/weights', (l['k'], l['k'], l['cin'], l['cout']))
        if l['bn']:
            out.append((w, (_Var(base + '/BatchNorm/gamma', (l['cout'],)), _Var(base + '/BatchNorm/beta', (l['cout'],)),
                            _Var(base + '/BatchNorm/moving_mean', (l['cout'],), False),
                            _Var(base + '/BatchNorm/moving_variance', (l['cout'],), False)), None))
        else:
            out.append((w, None, _Var(base + '/biases', (l['cout'],))))
    return out


def test_gradient_layout_follows_backward_order():
    from yolov3_tensorflow_amd import training, distributed
    lv = _layer_vars()
    order, offs, ends, total = training.gradient_layout(lv)
    assert len(order) == 75 + 72 * 2 + 3                         # kernels + gamma/beta + 3 detection biases
    assert 247.7e6 < total * 4 < 247.9e6            # 247.8 MB of fp32 gradients
    assert order[0].op_name == 'yolov3/yolov3_head/Conv_22/biases'         # produced first by backward
    assert order[-1].op_name == 'yolov3/darknet53_body/Conv/weights'        # ... and last
    assert all(o % 4 == 0 for o in offs.values())                # 16-byte aligned views
    assert sorted(ends) == list(range(75)) and ends[0] == total
    assert all(ends[i] > ends[i + 1] for i in range(74))         # monotone in production order
    edges = distributed.plan_buckets(sorted(ends.values()), 32 << 20)
    assert edges[-1] == total and all(b > a for a, b in zip(edges, edges[1:]))
    assert 7 <= len(edges) <= 9
    assert set(edges) <= set(ends.values())                      # bucket edges are layer edges
    # head-only fine-tuning (the reference's default update_part): 85.5 MB, backward stops at the head
    head = lambda v: v.trainable and v.op_name.startswith('yolov3/yolov3_head')
    order_h, _, ends_h, total_h = training.gradient_layout(lv, head)
    assert min(ends_h) == 52 and abs(total_h * 4 - 85.5e6) < 1e6
    assert all(v.op_name.startswith('yolov3/yolov3_head') for v in order_h)


def _worker(rank, world, init_file, out_dir):
    from yolov3_tensorflow_amd import training, distributed
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    lv = _layer_vars(class_num=3)[-12:]            # the last 12 layers are enough (keeps the CPU buffers small)
    order, offs, ends, total = training.gradient_layout(lv)
    flat = torch.zeros(total)
    ex = distributed.GradientExchange(flat, sorted(ends.values()), None, bucket_bytes=1 << 20)
    assert ex.world == world and ex.grad_scale == 1.0 / world
    issued_after_layer = {}
    for step in range(2):                           # two steps: begin() must reset the bucket cursor
        ex.begin()
        for li in range(len(lv) - 1, -1, -1):       # "backward": last layer first
            w, bnv, bias = lv[li]
            for v in ([w] + (list(bnv[:2]) if bnv is not None else [bias])):
                numel = 1
                for d in v.shape:
                    numel *= d
                view = flat[offs[v.op_name]:offs[v.op_name] + numel]
                view.copy_(torch.arange(numel, dtype=torch.float32) % 97 * (rank + 1) + li + step)
            ex.ready(ends[li])
            issued_after_layer[li] = len(ex.issued)
        ex.finish()
    mean = flat * ex.grad_scale
    # one-shot helper on a second buffer
    g = torch.full((1000,), float(rank + 1))
    distributed.all_reduce_mean_(g)
    # inference replicas: rank r takes images r, r+world, ... (no collective); the union must be the batch
    images = list(range(rank, 10, world))
    t = torch.tensor([float(rank == 0) * 3.0 + 1.0])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # bench.py: max over ranks of the elapsed time
    torch.save(dict(mean=mean, issued=ex.issued, after=issued_after_layer, edges=ex.edges, g=g, images=images,
                    tmax=float(t), total=total, ends=ends, offs=offs,
                    names=[v.op_name for v in order], shapes=[v.shape for v in order]),
               os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_gradient_exchange():
    d = tempfile.mkdtemp()
    mp.spawn(_worker, args=(2, os.path.join(d, 'init'), d), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(d, 'r0.pt')), torch.load(os.path.join(d, 'r1.pt'))
    assert torch.equal(r0['mean'], r1['mean'])                     # every rank holds the same averaged gradients
    # value check: rank r wrote (arange % 97) * (r + 1) + li + step (step = 1 in the last round) -> mean = 1.5 * a + li + 1
    lv_names, total = r0['names'], r0['total']
    want = torch.zeros(total)
    ends = r0['ends']
    layer_of = {}
    for li, e in ends.items():
        layer_of[e] = li
    starts = sorted(ends.values())
    for name, shape in zip(lv_names, r0['shapes']):
        numel = 1
        for s in shape:
            numel *= s
        off = r0['offs'][name]
        li = layer_of[min(e for e in starts if e > off)]
        want[off:off + numel] = torch.arange(numel, dtype=torch.float32) % 97 * 1.5 + li + 1
    assert torch.equal(r0['mean'], want)
    # buckets: contiguous cover of the buffer, more than one, issued in production order ...
    issued = r0['issued']
    assert len(issued) >= 3 and issued[0][0] == 0 and issued[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(issued, issued[1:]))
    # ... and overlapped: the first bucket went out before "backward" reached the first layer
    after = r0['after']
    assert after[min(after) + 1] >= 1, 'no bucket was issued before the last layer finished: nothing overlaps'
    assert after[min(after)] == len(issued)
    assert torch.equal(r0['g'], torch.full((1000,), 1.5)) and torch.equal(r1['g'], r0['g'])
    assert sorted(r0['images'] + r1['images']) == list(range(10)) and not set(r0['images']) & set(r1['images'])
    assert r0['tmax'] == r1['tmax'] == 4.0


def test_trainer_is_single_process_without_a_process_group():
    from yolov3_tensorflow_amd import training, distributed
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    opt = config_optimizer('momentum', 1e-3)
    assert isinstance(opt, training.Optimizer) and opt.kind == 'momentum' and opt.momentum == 0.9
    assert config_optimizer('rmsprop', 1e-3).epsilon == 1e-10 and config_optimizer('adam', 1e-3).epsilon == 1e-8
    assert opt.lr_at(5.0) == 1e-3
    assert training.Optimizer('sgd', lambda step: 0.1 * step).lr_at(3.0) == pytest.approx(0.3)
    assert not dist.is_initialized()
    flat = torch.arange(10, dtype=torch.float32)
    ex = distributed.GradientExchange(flat, [4, 10], None, bucket_bytes=16)
    ex.begin(); ex.ready(4); ex.finish()
    assert ex.world == 1 and ex.issued == [(0, 4), (4, 10)] and torch.equal(flat, torch.arange(10, dtype=torch.float32))


def test_bench_self_launches_ranks_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus 2` without a launcher must re-execute itself under torch.distributed.run (one rank per
    GPU) instead of exiting: on this GPU-less box the dry run prints the launch command and each rank's environment."""
    env = dict(os.environ, Y3_BENCH_DRY_RUN='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode()
    assert out.returncode == 0, text
    assert 'torch.distributed.run' in text and '--nproc-per-node 2' in text.replace('=', ' ')
    assert 'dry-run rank 0/2' in text and 'dry-run rank 1/2' in text
