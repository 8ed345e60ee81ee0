"""RCCL on the one GPU a box has: the train step through training.Trainer(process_group=WORLD) on a
world_size-1 `nccl` group launched under torch.distributed.run, with the bucket all-reduces FORCED (a one-rank group
would skip them).  This is the only RCCL evidence obtainable without a second GPU; the value semantics of two ranks are
covered by the gloo tests (tests/test_distributed_cpu.py, tests/test_data_parallel_gpu.py), the scaling curve by the
driver's 8-GPU run.

  (a) buckets are issued (asynchronous all-reduce works in flight) before backward has finished launching;
  (b) with ReduceOp.SUM the step equals the no-group step bit for bit (variables AND the flat gradient buffer);
  (c) stream ordering: with a pre-multiplied sum (factor 2) a one-rank all-reduce DOUBLES its bucket, so the result
      equals a no-group step whose clip/update kernel scales the gradients by 2 only if every all-reduce ran after the
      kernels that wrote its bucket and before mt_prepare_kernel read it — an ordering mistake between the process
      group's stream and the y3 context stream would leave a bucket undoubled (or doubled too late).
"""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_train_step_over_a_one_rank_rccl_group():
    out = tempfile.mkdtemp()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'workers', 'rccl_one_rank.py'), out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors='replace')[-4000:]
    with open(os.path.join(out, 'meta.json')) as f:
        meta = json.load(f)
    assert meta['backend'] == 'nccl' and meta['world'] == 1
    m = meta['sum']
    # (a) whole-model gradients (248 MB) in 16 MiB buckets: most of them go out while backward is still running
    assert m['buckets'] >= 8 and m['issued_before_end'] >= m['buckets'] - 1, m
    assert m['works_in_flight'] == m['issued_before_end'], m
    # (b) SUM over one rank is the identity: bit-exact against the no-group step
    a, b = np.load(os.path.join(out, 'sum_group.npz')), np.load(os.path.join(out, 'sum_local.npz'))
    assert set(a.files) == set(b.files) and len(a.files) > 300
    for k in a.files:
        assert np.array_equal(a[k], b[k]), 'SUM all-reduce over one rank changed %s' % k
    # (c) the ordering check
    if not meta['premul']:
        pytest.fail('this RCCL build rejects the pre-multiplied sum (%s): the stream-ordering check did not run'
                    % meta.get('premul_error'))
    a, b = np.load(os.path.join(out, 'premul_group.npz')), np.load(os.path.join(out, 'premul_local.npz'))
    flat_sum = np.load(os.path.join(out, 'sum_local.npz'))['flat']
    moved = 0
    for k in a.files:
        if k == 'flat':
            continue
        assert np.allclose(a[k], b[k], rtol=0, atol=1e-7 * max(1.0, float(np.abs(b[k]).max()))), \
            'premul all-reduce vs grad_scale=2: %s differs by %.3e' % (k, float(np.abs(a[k] - b[k]).max()))
        moved += int(not np.array_equal(b[k], np.load(os.path.join(out, 'sum_local.npz'))[k]))
    assert moved > 200, 'the factor-2 step must differ from the plain step for the trained variables (%d did)' % moved
    assert np.isfinite(flat_sum).all()
