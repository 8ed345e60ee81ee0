# coding: utf-8
"""The training / validation feeder (reference train.py:34-52 + utils/data_utils.py:179-224):

    tf.data.TextLineDataset -> shuffle -> batch -> map(py_func(get_batch_data), num_parallel_calls=10) -> prefetch(5)

rebuilt for one process per GPU with the device-resident target assignment of this package:

  * `num_threads` workers (reference: args.num_threads = 10) decode, augment and resize ONE image each
    (utils.data_utils.parse_sample).  They are worker PROCESSES by default: the numpy restatements of the OpenCV
    resizes and of the colour jitter hold the GIL for much of their time, and 16 threads measured 274 images/s where the
    train step wants 530 (tests/test_feeder_gpu.py); the children never touch the device.  The workers come from a
    `forkserver` (one pool per process and worker count, shared by the training and the validation feeder): forking the
    training process itself copies every PINNED host page eagerly - measured 93 s for four workers once 8 GB were
    pinned, against 0.4 s in a fresh process (tools/feeder_diag.py) - and the feeder is what pins them.  (As with any
    multiprocessing start method but fork, a SCRIPT that builds a Feeder needs the usual `if __name__ == '__main__':`
    guard: the workers import the main module.)  A coordinator thread assembles
    whole batches in PINNED host buffers and copies them to the device on a SIDE stream; a bounded queue of `prefetch` batches
    (reference: prefetech_buffer = 5) decouples it from the train step, so decode / resize / H2D of batch i+1.. overlap the
    step on batch i;
  * the consumer makes its compute stream wait for the copy's event (no host synchronisation) and runs `y3_process_box`
    for the whole batch on the device (utils.data_utils.process_box_batch: bit-exact against the reference's
    process_box), so the three y_true tensors (3.6 MB per 416x416 image - more than the image itself) never cross PCIe;
  * multi-scale: a new size every `interval` batches, drawn exactly like get_batch_data (random.seed(count // interval)
    over range(10, 20) * 32 = 320 .. 608); mix-up pairing per batch; every random draw comes from generators seeded by
    (seed, epoch, batch, sample), so a run is reproducible whatever the thread interleaving (the reference's is not);
  * data parallel: rank r of `world` takes samples r::world of every global batch (same shuffle on every rank).
"""
from __future__ import division, print_function

import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _worker_sample(job):
    """One sample in a worker (process or thread): job = (line or mix-up pair, [w, h], mode, letterbox, rng key)."""
    from .utils.data_utils import parse_sample
    line, size, mode, letterbox, key = job
    return parse_sample(line, size, mode, letterbox, rng=np.random.RandomState(key % (2 ** 31)), prng=random.Random(key),
                        as_uint8=True)


_POOLS = {}
_POOLS_LOCK = threading.Lock()


def _shared_process_pool(workers):
    import atexit
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    with _POOLS_LOCK:
        pool = _POOLS.get(workers)
        if pool is None:
            ctx = multiprocessing.get_context('forkserver')
            ctx.set_forkserver_preload(['yolov3_tensorflow_amd.utils.data_utils', 'yolov3_tensorflow_amd.utils.data_aug'])
            pool = _POOLS[workers] = ProcessPoolExecutor(workers, mp_context=ctx)
            if len(_POOLS) == 1:
                atexit.register(_shutdown_pools)
        return pool


def _shutdown_pools():
    with _POOLS_LOCK:
        for pool in _POOLS.values():
            pool.shutdown(wait=False, cancel_futures=True)
        _POOLS.clear()


class Batch(object):
    """image_ids (list), images [n,h,w,3] float32 device tensor, y_true (three device tensors), img_size [w, h]."""
    __slots__ = ('image_ids', 'images', 'y_true', 'img_size', 'boxes', 'labels', 'counts')


class Feeder(object):
    def __init__(self, lines, batch_size, class_num, img_size, anchors, mode='train', multi_scale=False, use_mix_up=False,
                 letterbox_resize=True, num_threads=10, prefetch=5, shuffle=None, seed=0, rank=0, world=1, interval=10,
                 device=None, drop_remainder=False, backend='process'):
        self.lines = [l for l in lines if (l.strip() if isinstance(l, str) else l)]
        self.batch_size, self.class_num = int(batch_size), int(class_num)
        self.img_size, self.anchors = list(img_size), np.asarray(anchors, np.float32).reshape(9, 2)
        self.mode = mode
        self.multi_scale = bool(multi_scale) and mode == 'train'
        self.use_mix_up = bool(use_mix_up) and mode == 'train'
        self.letterbox = bool(letterbox_resize)
        self.num_threads, self.prefetch = max(1, int(num_threads)), max(1, int(prefetch))
        self.shuffle = (mode == 'train') if shuffle is None else bool(shuffle)
        self.seed, self.rank, self.world, self.interval = int(seed), int(rank), int(world), int(interval)
        self.device = device
        self.drop_remainder = drop_remainder
        self.batches_served = 0          # the reference's iter_cnt: counts batches over epochs (multi-scale schedule)
        if backend not in ('process', 'thread'):
            raise ValueError("backend must be 'process' or 'thread'")
        self.backend = backend
        self._pool = None

    def _executor(self):
        """The worker pool: one per process and worker count, created on first use and shared by every Feeder (see the
        module docstring for why the workers are not forked from this process)."""
        if self.backend == 'thread':
            if self._pool is None:
                self._pool = ThreadPoolExecutor(self.num_threads)
            return self._pool
        return _shared_process_pool(self.num_threads)

    def close(self):
        if self._pool is not None:          # (only thread pools are owned by the feeder)
            self._pool.shutdown(wait=False, cancel_futures=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass

    def __len__(self):
        n = len(self.lines)
        return n // self.batch_size if self.drop_remainder else (n + self.batch_size - 1) // self.batch_size

    # ---- host side ------------------------------------------------------------------------------------------------
    def _plan(self, epoch):
        """[(batch number within the epoch, img_size, [line or mix-up pair, ...] of THIS rank)] for one epoch."""
        order = list(range(len(self.lines)))
        if self.shuffle:
            random.Random(self.seed * 1000003 + epoch).shuffle(order)      # same order on every rank
        plan = []
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            size = list(self.img_size)
            if self.multi_scale:
                from .utils.data_utils import multi_scale_size
                size = multi_scale_size(self.batches_served + b, self.interval)
            lines = [self.lines[i] for i in idx]
            if self.use_mix_up:
                from .utils.data_utils import mix_up_lines
                key = (self.seed * 1000003 + epoch) * 100003 + b
                lines = mix_up_lines(lines, rng=np.random.RandomState(key % (2 ** 31)), prng=random.Random(key))
            mine = lines[self.rank::self.world] or lines[:1]
            plan.append((b, size, mine))
        return plan

    def _job(self, epoch, b, j, line, size):
        key = ((self.seed * 1000003 + epoch) * 100003 + b) * 1009 + j * self.world + self.rank
        return (line, size, self.mode, self.letterbox, key)

    def epoch(self, epoch=0):
        """Iterate over one epoch: yields Batch objects whose tensors live on the device."""
        import torch
        from . import framework as fw
        from .utils.data_utils import collate, process_box_batch
        dev = torch.device(self.device) if self.device is not None else fw.default_device()
        plan = self._plan(epoch)
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        copy_stream = torch.cuda.Stream(device=dev)

        pool = self._executor()

        def produce():
            try:
                # keep up to `prefetch` batches of decode jobs in flight
                pending = []
                it = iter(plan)
                exhausted = False
                while not stop.is_set():
                    while not exhausted and len(pending) < self.prefetch:
                        try:
                            b, size, lines = next(it)
                        except StopIteration:
                            exhausted = True
                            break
                        futs = [pool.submit(_worker_sample, self._job(epoch, b, j, line, size))
                                for j, line in enumerate(lines)]
                        pending.append((b, size, futs))
                    if not pending:
                        break
                    b, size, futs = pending.pop(0)
                    samples = [f.result() for f in futs]
                    n = len(samples)
                    pinned = torch.empty((n, size[1], size[0], 3), dtype=torch.float32).pin_memory()
                    ids, _, boxes, labels, counts = collate(samples, out_images=pinned.numpy())
                    with torch.cuda.stream(copy_stream):
                        images = pinned.to(dev, non_blocking=True)
                        bx = torch.from_numpy(boxes).pin_memory().to(dev, non_blocking=True)
                        lb = torch.from_numpy(labels.astype(np.int32)).pin_memory().to(dev, non_blocking=True)
                        ct = torch.from_numpy(counts.astype(np.int32)).pin_memory().to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    item = (ids, size, images, bx, lb, ct, ev, pinned)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as e:       # noqa: BLE001 - handed to the consumer
                q.put(e)

        th = threading.Thread(target=produce, name='y3-feeder', daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                ids, size, images, bx, lb, ct, ev, pinned = item
                torch.cuda.current_stream(dev).wait_event(ev)         # device-side ordering only
                for t in (images, bx, lb, ct):
                    t.record_stream(torch.cuda.current_stream(dev))
                out = Batch()
                out.image_ids, out.images, out.img_size = ids, images, size
                out.boxes, out.labels, out.counts = bx, lb, ct
                out.y_true = process_box_batch(bx, lb, ct, size, self.class_num, self.anchors)
                self.batches_served += 1
                yield out
        finally:
            stop.set()
            th.join(timeout=5.0)

    def __iter__(self):
        return self.epoch(0)
