"""The feeder on the GPU box (ref: train.py:34-52, utils/data_utils.py:118-224): batches arrive
device-resident with the targets assigned by y3_process_box, and a train loop fed by it runs at the speed of the same loop
on resident tensors - decode / augmentation / resize / H2D of the next batches overlap the step."""
import time

import numpy as np
import pytest
import torch

from conftest import COCO_ANCHORS

pytestmark = pytest.mark.gpu


def _write_set(tmp_path, n, classes=80, seed=5):
    from PIL import Image
    rng = np.random.RandomState(seed)
    lines = []
    for i in range(n):
        w, h = int(rng.randint(280, 360)), int(rng.randint(200, 280))
        path = str(tmp_path / ('img_%d.jpg' % i))
        base = rng.randint(0, 256, (h // 8, w // 8, 3)).astype(np.uint8)
        Image.fromarray(np.repeat(np.repeat(base, 8, 0), 8, 1)[:h, :w]).save(path, quality=85)
        k = int(rng.randint(1, 5))
        parts = ['%d' % i, path, '%d' % w, '%d' % h]
        for _ in range(k):
            x0, y0 = rng.uniform(0, w * 0.5), rng.uniform(0, h * 0.5)
            parts += ['%d' % rng.randint(0, classes), '%.1f' % x0, '%.1f' % y0, '%.1f' % (x0 + rng.uniform(20, w * 0.45)),
                      '%.1f' % (y0 + rng.uniform(20, h * 0.45))]
        lines.append(' '.join(parts))
    return lines


def test_feeder_batches_are_device_resident_and_targets_match_the_oracle(tmp_path):
    from yolov3_tensorflow_amd.feeder import Feeder
    from oracle import train_ref
    lines = _write_set(tmp_path, 21)
    f = Feeder(lines, 8, 80, [416, 416], COCO_ANCHORS, mode='train', multi_scale=True, use_mix_up=True, num_threads=6,
               prefetch=3, seed=1)
    seen = 0
    for batch in f.epoch(0):
        n = len(batch.image_ids)
        w, h = batch.img_size
        assert batch.images.is_cuda and tuple(batch.images.shape) == (n, h, w, 3) and batch.images.dtype == torch.float32
        assert 0.0 <= float(batch.images.min()) and float(batch.images.max()) <= 1.0
        assert [tuple(y.shape) for y in batch.y_true] == [(n, h // s, w // s, 3, 86) for s in (32, 16, 8)]
        bx, lb, ct = batch.boxes.cpu().numpy(), batch.labels.cpu().numpy(), batch.counts.cpu().numpy()
        for i in range(n):
            want = train_ref.process_box(bx[i, :ct[i]], lb[i, :ct[i]], [w, h], 80, COCO_ANCHORS)       # box i <-> labels[i]
            for got, ref in zip(batch.y_true, want):
                np.testing.assert_array_equal(got[i].cpu().numpy(), ref)
        seen += n
    assert seen == 21 and f.batches_served == 3
    # the same epoch again: identical batches (every draw is seeded by (seed, epoch, batch, sample))
    f2 = Feeder(lines, 8, 80, [416, 416], COCO_ANCHORS, mode='train', multi_scale=True, use_mix_up=True, num_threads=3,
                prefetch=2, seed=1)
    a = next(iter(f2.epoch(0)))
    b = next(iter(Feeder(lines, 8, 80, [416, 416], COCO_ANCHORS, mode='train', multi_scale=True, use_mix_up=True,
                         num_threads=9, prefetch=5, seed=1).epoch(0)))
    assert torch.equal(a.images, b.images) and all(torch.equal(x, y) for x, y in zip(a.y_true, b.y_true))


def test_train_loop_fed_by_the_feeder_keeps_the_resident_step_time(tmp_path, isolated_graph):
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.feeder import Feeder
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    # six workers: at bs=16 the step is ~29 ms of which the host's ~1,100 kernel launches are a good part, and the pod's cgroup
    # grants 16 cores - 10 to 32 busy workers take the launching thread's core away (fed 32-36 ms against 29.5 resident, one
    # call, same box; 29.7 with six, who still feed a batch in 8-11 ms).  At bs=64 the reference's ten workers cost 1 %
    # (bench.py: c4.fed).
    bs, steps, workers = 16, 12, 6
    lines = _write_set(tmp_path, bs * (steps + 8), seed=9)
    y3.reset_default_graph()
    model = y3.yolov3(80, COCO_ANCHORS, batch_norm_decay=0.99)
    model.compute_dtype = 'f32_wino'
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 32, 32, 3))
        trainer = training.Trainer(model, config_optimizer('momentum', 1e-4))
        feeder = Feeder(lines, bs, 80, [416, 416], COCO_ANCHORS, mode='train', use_mix_up=True, num_threads=workers,
                        prefetch=5, seed=2)
        it = feeder.epoch(0)
        first = next(it)
        it.close()
        for _ in range(2):                                   # warm-up (allocations, weight packing) on a resident batch
            trainer.step(first.images, first.y_true)

        def measure(epoch):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                trainer.step(first.images, first.y_true)
            torch.cuda.synchronize()
            resident = (time.perf_counter() - t0) / steps
            it = feeder.epoch(epoch)
            next(it)
            t0 = time.perf_counter()
            done = 0
            for batch in it:
                trainer.step(batch.images, batch.y_true)
                done += 1
                if done == steps:
                    break
            torch.cuda.synchronize()
            fed = (time.perf_counter() - t0) / steps
            it.close()
            # the feeder on its own on THIS host (no step to hide under): what a batch costs when the host is the slower stage
            it = feeder.epoch(epoch)
            for _ in range(6):                               # what was prefetched before the clock starts does not count
                next(it)
            t0 = time.perf_counter()
            for _ in range(steps):
                next(it)
            torch.cuda.synchronize()
            alone = (time.perf_counter() - t0) / steps
            it.close()
            return done, resident, fed, alone

        # The two stages overlap: the loop runs at the pace of the slower one (on the GPU boxes seen so far that is the
        # step: 22-28 ms against 11-16 ms of feeding).  The boxes share a 256-core host with other jobs (load average 12
        # and more): a burst there must not turn a property of the code into a failure - up to three attempts.
        for attempt in range(3):
            done, resident, fed, alone = measure(attempt)
            print('train step bs=%d @416: %.1f ms on resident tensors, %.1f ms fed by the feeder (%d threads, prefetch %d): '
                  '%.0f images/s decoded, augmented, resized and uploaded under the steps; the feeder alone: %.1f ms per batch'
                  % (bs, resident * 1e3, fed * 1e3, workers, 5, bs / fed, alone * 1e3))
            assert done == steps
            if fed <= 1.15 * max(resident, alone) + 4e-3:
                break
    feeder.close()
    # (absolute part: what a fed step does on top of a resident one whatever its length - the side-stream upload of a bs=16 batch,
    # the target-assignment launches, the hand-over - 2 ms while the step took 28 ms; the step is 25.9 ms since round 6 and the
    # fed loop measured 29.7 alone on a box and failed three attempts inside the full suite: 4 ms)
    assert fed <= 1.15 * max(resident, alone) + 4e-3, (fed, resident, alone)
    # ... and the feeder itself has a budget of its own (the overlap check above moves with `alone`: a feeder that regressed to
    # 100 ms per batch would still "keep its own pace"): a bs=16 batch - decode, augmentation, resize, upload, target assignment -
    # in at most 40 ms on the 16 cores of these pods (measured 11-16 ms), whatever the step costs
    assert alone <= 40e-3, 'the feeder alone needs %.1f ms per bs=%d batch' % (alone * 1e3, bs)


def test_process_backed_feeder_fills_shared_pinned_buffers_and_equals_the_thread_backed_one(tmp_path):
    """backend='process': worker processes write their float32 slots into batch buffers in /dev/shm that the parent has
    page-locked for the device (hipHostRegister) - the arrangement that scales past one interpreter's GIL (3,320 images/s
    with 32 workers against ~1,000 for threads, profiles/r03_feeder_rate.txt).  Same seed, same batches as the thread-backed
    feeder, and nothing is left behind in /dev/shm."""
    import glob
    import os
    from yolov3_tensorflow_amd.feeder import Feeder
    lines = _write_set(tmp_path, 24, seed=4)
    first = {}
    for backend in ('thread', 'process'):
        f = Feeder(lines, 8, 80, [416, 416], COCO_ANCHORS, mode='train', multi_scale=True, use_mix_up=True, num_threads=4,
                   prefetch=2, seed=6, backend=backend)
        it = f.epoch(0)
        batches = [next(it) for _ in range(2)]
        torch.cuda.synchronize()
        first[backend] = [(b.image_ids, b.images.cpu(), [y.cpu() for y in b.y_true]) for b in batches]
        it.close()
        f.close()
    for (ids_t, img_t, y_t), (ids_p, img_p, y_p) in zip(first['thread'], first['process']):
        assert ids_t == ids_p and torch.equal(img_t, img_p) and all(torch.equal(a, b) for a, b in zip(y_t, y_p))
    assert not glob.glob('/dev/shm/y3feed_%d_*' % os.getpid())


def test_closing_an_epoch_early_releases_the_producer_at_once(tmp_path):
    """with the queue full (the feeder runs ahead of its consumer) an early `close()` of the epoch left the
    producer thread in a blocking `q.put` for good - a 5 s stall per close, and the prefetched device batches and pinned
    buffers stayed alive.  Now every hand-over gives up once the consumer has stopped, the queue is drained and the
    thread is gone when `close()` returns."""
    import threading
    from yolov3_tensorflow_amd.feeder import Feeder
    lines = _write_set(tmp_path, 48)
    for backend in ('thread', 'process'):
        f = Feeder(lines, 4, 80, [224, 224], COCO_ANCHORS, mode='train', use_mix_up=True, num_threads=4, prefetch=4, seed=2,
                   backend=backend)
        it = f.epoch(0)
        next(it)
        time.sleep(1.0)                       # the producer fills the queue and blocks on the next hand-over
        t0 = time.perf_counter()
        it.close()
        dt = time.perf_counter() - t0
        f.close()
        assert dt < 2.5, '%s backend: closing the epoch took %.1f s' % (backend, dt)
        assert not [t for t in threading.enumerate() if t.name == 'y3-feeder' and t.is_alive()], backend
