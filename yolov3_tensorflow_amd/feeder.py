# coding: utf-8
"""The training / validation feeder (reference train.py:34-52 + utils/data_utils.py:179-224):

    tf.data.TextLineDataset -> shuffle -> batch -> map(py_func(get_batch_data), num_parallel_calls=10) -> prefetch(5)

rebuilt for one process per GPU with the device-resident target assignment of this package:

  * `num_threads` workers (reference: args.num_threads = 10) decode, augment and resize ONE image each
    (utils.data_utils.parse_sample).  The pixel work - blend, colour jitter, crop, resize, pad, flip, /255 - and the crop
    search run in liby3feed.so (include/yolo355_feed.h: native code where the reference's is OpenCV), which releases the
    GIL, as PIL's JPEG decoder does; a worker holds the GIL for a little under a millisecond per image (which levels one
    process off at 1,000-1,200 images/s: profiles/r03_feeder_rate.txt).  So the workers are
    THREADS by default (backend='thread') and each writes its float32 image straight into its slot of the batch's pinned
    host buffer: no pickling, no second pass.  backend='process' is the arrangement that scales past one interpreter's
    GIL: worker processes from a `forkserver` (one pool per process and worker count, shared by the training and the
    validation feeder) write their float32 slots into batch buffers that live in /dev/shm and that the parent has
    page-locked for the device (hipHostRegister), so only boxes and labels travel through the pipes: 3,320 images/s with
    32 workers; it is also the default for the numpy / Pillow pixel path (Y3_FEED_NATIVE=0), which holds the GIL most of the
    time.  (Without /dev/shm the workers hand back 8-bit images that the coordinator divides into a pinned buffer: the
    round-2 arrangement.)  Why a forkserver: forking the
    training process itself copies every PINNED host page eagerly - measured 93 s for four workers once 8 GB were pinned,
    against 0.4 s in a fresh process (tools/feeder_diag.py) - and the feeder is what pins them.  (As with any
    multiprocessing start method but fork, a SCRIPT that builds a process-backed Feeder needs the usual
    `if __name__ == '__main__':` guard.)
  * a coordinator thread keeps `prefetch` batches of jobs in flight, and copies each finished batch to the device on a
    SIDE stream; a bounded queue of `prefetch` batches (reference: prefetech_buffer = 5) decouples it from the train step,
    so decode / resize / H2D of batch i+1.. overlap the step on batch i.  The pinned buffers are recycled once their copy
    has completed (pinning 130 MB per batch afresh costs more than filling it);
  * pixels='gpu': the workers only decode and draw (parse_sample(defer=True)); the coordinator plans the batch's pixel
    work (liby3feed.so: y3f_plan_batch, native threads), uploads ONE blob of 8-bit sources and tables and runs y3_feed_run
    on the side stream (feed_device.DevicePixels) - the same bytes as the host path, 5 of the 5.9 ms a core spends per
    image moved to three launches of a few hundred microseconds per batch;
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
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _worker_sample(job, out=None, defer=False):
    """One sample in a worker: job = (line or mix-up pair, [w, h], mode, letterbox, rng key).  A thread is handed `out`, its
    float32 slot of the batch buffer; a process returns the 8-bit image (a quarter of the bytes to pickle); `defer`: the
    decoded sources and the pixel JOB come back, the pixels are left to the device."""
    from .utils.data_utils import parse_sample
    line, size, mode, letterbox, key = job
    return parse_sample(line, size, mode, letterbox, rng=np.random.RandomState(key % (2 ** 31)), prng=random.Random(key),
                        as_uint8=out is None and not defer, out=out, defer=defer)


_ATTACHED = {}        # (in a worker process) path of a shared batch buffer -> its mapping


def _worker_sample_shared(job, path, shape, index):
    """One sample in a worker PROCESS, written as float32 straight into slot `index` of the batch buffer the parent keeps in
    shared memory at `path` ([n, h, w, 3]); only the boxes and labels travel back through the pipe."""
    arr = _ATTACHED.get(path)
    if arr is None:
        if len(_ATTACHED) >= 32:            # the parent recycles a handful of buffers; anything older is gone
            _ATTACHED.clear()
        arr = _ATTACHED[path] = np.memmap(path, dtype=np.float32, mode='r+', shape=tuple(shape))
    idx, _, boxes, labels = _worker_sample(job, out=arr[index])
    return idx, None, boxes, labels


class _SharedBuffers(object):
    """Batch buffers for worker processes: files in /dev/shm mapped here and in the workers, page-locked for the device
    (hipHostRegister) so that the upload reads them like any pinned buffer; recycled like _PinnedBuffers.  Where shared memory
    or the registration is not available, `take` returns None and the feeder falls back to pickled 8-bit images."""

    _live = []              # instances with files in /dev/shm: closed at interpreter exit at the latest

    def __init__(self):
        self.busy = []          # (event, entry)
        self.entries = []       # every live entry: dict(path, map, array, tensor, registered)
        self.broken = False
        if not _SharedBuffers._live:
            import atexit
            atexit.register(_SharedBuffers._close_all)
        _SharedBuffers._live.append(self)

    @staticmethod
    def _close_all():
        for inst in list(_SharedBuffers._live):
            inst.close()

    def _create(self, shape):
        import mmap
        import os
        import tempfile
        import torch
        nbytes = int(np.prod(shape)) * 4
        fd, path = tempfile.mkstemp(prefix='y3feed_%d_' % os.getpid(), dir='/dev/shm')
        try:
            os.ftruncate(fd, nbytes)
            mapping = mmap.mmap(fd, nbytes)
        finally:
            os.close(fd)
        array = np.frombuffer(mapping, np.float32).reshape(shape)
        tensor = torch.from_numpy(array)
        registered = False
        have_device = torch.cuda.is_available()
        if have_device:
            try:
                registered = int(torch.cuda.cudart().cudaHostRegister(tensor.data_ptr(), nbytes, 0)) == 0
            except Exception:       # noqa: BLE001 - no such entry point
                registered = False
        if have_device and not registered:
            # The raw return code leaves HIP's per-thread last-error set (memlock ulimit, an unsupported /dev/shm mapping):
            # clear it, or the next launch check of PyTorch on this thread reports an unrelated "HIP error".  And stop
            # using shared buffers: an unpinned mapping would mean synchronous pageable copies - the pickled 8-bit path
            # (take() -> None) is the better fallback.
            try:
                import ctypes
                ctypes.CDLL('libamdhip64.so').hipGetLastError()
            except OSError:
                pass
            self.broken = True
        entry = dict(path=path, map=mapping, array=array, tensor=tensor, registered=registered)
        self.entries.append(entry)
        return entry

    def take(self, shape):
        if self.broken:
            return None
        shape = tuple(int(v) for v in shape)
        free = [i for i, (ev, _) in enumerate(self.busy) if ev.query()]
        for i in free:
            if tuple(self.busy[i][1]['array'].shape) == shape:
                return self.busy.pop(i)[1]
        for i in reversed(free[:-4]):
            self._release(self.busy.pop(i)[1])
        try:
            entry = self._create(shape)
        except (OSError, ValueError):
            self.broken = True
            return None
        if self.broken:                 # (the mapping could not be page-locked)
            self._release(entry)
            return None
        return entry

    def give(self, event, entry):
        self.busy.append((event, entry))

    def _release(self, entry):
        import os
        import torch
        if entry in self.entries:
            self.entries.remove(entry)
        try:
            if entry['registered']:
                torch.cuda.cudart().cudaHostUnregister(entry['tensor'].data_ptr())
        except Exception:       # noqa: BLE001
            pass
        entry['tensor'] = entry['array'] = None
        try:
            entry['map'].close()
        except (BufferError, ValueError):       # a view is still alive somewhere: the mapping goes with it
            pass
        try:
            os.unlink(entry['path'])
        except OSError:
            pass

    def close(self):
        self.busy = []
        for entry in list(self.entries):
            self._release(entry)
        if self in _SharedBuffers._live:
            _SharedBuffers._live.remove(self)


class _PinnedBuffers(object):
    """Pinned host batch buffers by shape, handed out again once the H2D copy that read them has completed."""

    def __init__(self):
        self.busy = []          # (event, tensor)

    def take(self, shape):
        import torch
        shape = tuple(int(v) for v in shape)
        free = [i for i, (ev, _) in enumerate(self.busy) if ev.query()]
        for i in free:
            if tuple(self.busy[i][1].shape) == shape:
                return self.busy.pop(i)[1]
        # none of this shape is free.  Multi-scale training moves from size to size: keep a handful of free buffers of
        # other shapes (the size may come back), unpin the rest
        for i in reversed(free[:-4]):
            del self.busy[i]
        return torch.empty(shape, dtype=torch.float32).pin_memory()

    def give(self, event, tensor):
        self.busy.append((event, tensor))


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
                 device=None, drop_remainder=False, backend=None, pixels=None):
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
        if backend is None:      # threads when the pixel work is native code that releases the GIL (module docstring)
            from . import feed_native
            backend = 'thread' if feed_native.enabled() else 'process'
        if backend not in ('process', 'thread'):
            raise ValueError("backend must be 'process' or 'thread'")
        self.backend = backend
        if pixels is None:      # on the device wherever the workers can hand decoded sources over (same bytes either way)
            from . import feed_native
            pixels = 'gpu' if backend == 'thread' and feed_native.enabled() else 'host'
        if pixels not in ('host', 'gpu'):
            raise ValueError("pixels must be 'host' or 'gpu'")
        if pixels == 'gpu':
            from . import feed_native
            if backend != 'thread' or not feed_native.enabled():
                raise ValueError("pixels='gpu' runs with the thread backend and the native job builder (Y3_FEED_NATIVE=1): "
                                 "the workers hand decoded sources to the coordinator")
        self.pixels = pixels
        self._pool = None
        self._side = None           # (device, side stream, DevicePixels or None)

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
        # one side stream (and, with pixels='gpu', one set of device tables + pinned blob buffers) per feeder and device, kept
        # over the epochs: framework.context() caches a y3_ctx per (device, stream)
        if self._side is None or self._side[0] != dev:
            self._side = (dev, torch.cuda.Stream(device=dev), None)
        copy_stream = self._side[1]

        pool = self._executor()

        on_device = self.pixels == 'gpu'
        device_pixels = None
        if on_device:
            if self._side[2] is None:
                from .feed_device import DevicePixels
                self._side = (dev, copy_stream, DevicePixels(dev))
            device_pixels = self._side[2]
        in_place = self.backend == 'thread'
        buffers = _PinnedBuffers()
        shared = None if in_place else _SharedBuffers()

        def put(item):
            # every hand-over gives up as soon as the consumer has stopped: a blocking put on a full queue that nobody
            # drains any more would park this thread - and the device tensors / pinned buffers it holds - for good
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return
                except queue.Full:
                    continue

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
                        shape = (len(lines), size[1], size[0], 3)
                        jobs = [self._job(epoch, b, j, line, size) for j, line in enumerate(lines)]
                        entry = shared.take(shape) if shared is not None else None
                        if on_device:               # worker threads decode and draw; the pixels are left to the device
                            pinned, owner = None, None
                            futs = [pool.submit(_worker_sample, job, None, True) for job in jobs]
                        elif entry is not None:     # worker processes fill the shared, page-locked batch buffer
                            pinned, owner = entry['tensor'], (shared, entry)
                            futs = [pool.submit(_worker_sample_shared, job, entry['path'], shape, j)
                                    for j, job in enumerate(jobs)]
                        else:
                            pinned = buffers.take(shape)
                            owner = (buffers, pinned)
                            slots = pinned.numpy()
                            if in_place:            # worker threads fill the pinned batch buffer
                                futs = [pool.submit(_worker_sample, job, slots[j]) for j, job in enumerate(jobs)]
                            else:                   # worker processes hand back 8-bit images
                                futs = [pool.submit(_worker_sample, job) for job in jobs]
                        pending.append((b, size, futs, pinned, owner))
                    if not pending:
                        break
                    b, size, futs, pinned, owner = pending.pop(0)
                    samples = [f.result() for f in futs]
                    if on_device:
                        ids, _, boxes, labels, counts = collate(samples, with_images=False)
                    else:
                        slots = pinned.numpy()
                        samples = [(s[0], slots[j] if s[1] is None else s[1], s[2], s[3]) for j, s in enumerate(samples)]
                        ids, _, boxes, labels, counts = collate(samples, out_images=slots)
                    with torch.cuda.stream(copy_stream):
                        if on_device:
                            images = device_pixels.run([s[1] for s in samples], threads=min(self.num_threads, 8))
                        else:
                            images = pinned.to(dev, non_blocking=True)
                        bx = torch.from_numpy(boxes).pin_memory().to(dev, non_blocking=True)
                        lb = torch.from_numpy(labels.astype(np.int32)).pin_memory().to(dev, non_blocking=True)
                        ct = torch.from_numpy(counts.astype(np.int32)).pin_memory().to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    if owner is not None:
                        owner[0].give(ev, owner[1])
                    item = (ids, size, images, bx, lb, ct, ev)
                    put(item)
                put(None)
            except BaseException as e:       # noqa: BLE001 - handed to the consumer
                put(e)

        th = threading.Thread(target=produce, name='y3-feeder', daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                ids, size, images, bx, lb, ct, ev = item
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
            deadline = time.monotonic() + 30.0
            while th.is_alive() and time.monotonic() < deadline:      # drain what is queued so that nothing the producer
                try:                                                  # holds outlives the epoch (bounded: a worker that
                    q.get_nowait()                                    # never returns must not hang the caller's close())
                except queue.Empty:
                    th.join(timeout=0.05)
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            copy_stream.synchronize()
            if shared is not None:
                shared.close()

    def __iter__(self):
        return self.epoch(0)
