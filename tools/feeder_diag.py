#!/usr/bin/env python
"""Why the feeder's worker PROCESSES (backend='process'; the default backend is threads over liby3feed.so) come from a
forkserver: time to start four workers and get one batch of samples back, in a fresh process and after the process has
pinned 8 GB of host memory (what a training process does: the feeder's own batches are pinned), with workers FORKED from
this process against workers from the forkserver pool.

    python tools/feeder_diag.py          (needs a GPU for the pinned allocations)

Measured on one MI355X box (round 3): fork 0.4 s fresh / 93 s after 8 GB pinned (the kernel copies pinned pages eagerly
at fork); forkserver ~2 s the first time (server start + imports), ~0.1 s afterwards, whatever is pinned.
"""
import multiprocessing
import os
import pathlib
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def one_batch(pool, lines):
    from yolov3_tensorflow_amd import feeder
    t0 = time.perf_counter()
    futs = [pool.submit(feeder._worker_sample, (l, [160, 160], 'train', True, 100 + i)) for i, l in enumerate(lines)]
    [f.result() for f in futs]
    return time.perf_counter() - t0


def main():
    import torch
    from test_feeder_gpu import _write_set
    from yolov3_tensorflow_amd import feeder
    lines = _write_set(pathlib.Path(tempfile.mkdtemp()), 8)
    for tag in ('fresh process', 'after pinning 8 GB'):
        if tag != 'fresh process':
            keep = [torch.empty(1 << 28, dtype=torch.float32).pin_memory() for _ in range(8)]      # noqa: F841
        forked = ProcessPoolExecutor(4, mp_context=multiprocessing.get_context('fork'))
        print('%-20s fork:       first batch %.2f s, next %.3f s' % (tag, one_batch(forked, lines), one_batch(forked, lines)),
              flush=True)
        forked.shutdown(wait=False, cancel_futures=True)
        feeder._shutdown_pools()
        pool = feeder._shared_process_pool(4)
        print('%-20s forkserver: first batch %.2f s, next %.3f s' % (tag, one_batch(pool, lines), one_batch(pool, lines)),
              flush=True)


if __name__ == '__main__':
    main()
