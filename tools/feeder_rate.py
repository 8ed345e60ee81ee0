#!/usr/bin/env python
"""Standalone throughput of the feeder: images/s decoded, augmented, resized and copied to the device with nothing
consuming them but this loop - how far the host side is from what the train step asks for (704 images/s per GPU at
416x416, bs=64: profiles/r03_bench_c4_n1.json).  Synthetic 640x480 JPEGs (COCO's usual size), 'train' mode with mix-up,
multi-scale off, batch 64.

    python tools/feeder_rate.py [--workers 8,16,32] [--backends thread,process] [--native 1,0] [--batches 12]
    python tools/feeder_rate.py --feeders 8 [--workers 10] [--backends thread]     # the host side of an 8-GPU node:
        eight feeder PROCESSES (one per rank, as train.py runs them) at once on this host, aggregate images/s against
        the 8 x 697 images/s eight train steps consume (VERDICT r3 #5b; every process uploads to cuda:0 here)
"""
import argparse
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326], np.float32)


def write_set(folder, n, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    lines = []
    for i in range(n):
        w, h = 640, 480
        small = rng.randint(0, 256, (h // 10, w // 10, 3)).astype(np.uint8)
        path = str(folder / ('img_%d.jpg' % i))
        Image.fromarray(small).resize((w, h), Image.BICUBIC).save(path, quality=90)
        parts = ['%d' % i, path, '%d' % w, '%d' % h]
        for _ in range(int(rng.randint(1, 8))):
            x0, y0 = rng.uniform(0, w * 0.5), rng.uniform(0, h * 0.5)
            parts += ['%d' % rng.randint(0, 80), '%.1f' % x0, '%.1f' % y0, '%.1f' % (x0 + rng.uniform(20, w * 0.45)),
                      '%.1f' % (y0 + rng.uniform(20, h * 0.45))]
        lines.append(' '.join(parts))
    return lines


def _one_feeder(rank, nfeed, lines, args, backend, workers, barrier, out):
    """One rank's feeder in its own process: warm up, meet the others, serve `batches` batches, report the window."""
    import torch
    from yolov3_tensorflow_amd.feeder import Feeder
    try:
        f = Feeder(lines, args.batch_size, 80, [416, 416], ANCHORS, mode='train', use_mix_up=True, num_threads=workers,
                   prefetch=5, seed=1 + rank, backend=backend, pixels=args.pixels.split(',')[0])
        if args.host_only:
            # the host half alone (decode, augmentation chain, resize into a float32 batch buffer, collate): no device call at
            # all, so eight of these on a one-GPU box measure the HOST of an eight-GPU node (with uploads, eight processes
            # time-slice the one device here and the figure is the slicing, not the host)
            from yolov3_tensorflow_amd import feeder as fd
            from yolov3_tensorflow_amd.utils.data_utils import collate
            pool, plan = f._executor(), f._plan(0)

            free = []           # batch buffers are recycled, as the feeder's pinned buffers are (no page faults per batch)

            def submit(entry):
                b, size, mine = entry
                shape = (len(mine), size[1], size[0], 3)
                slots = free.pop() if free and free[-1].shape == shape else np.empty(shape, np.float32)
                return slots, [pool.submit(fd._worker_sample, f._job(0, b, j, line, size), slots[j]) for j, line in enumerate(mine)]

            def finish(item):
                slots, futs = item
                samples = [x.result() for x in futs]
                collate([(s_[0], slots[j] if s_[1] is None else s_[1], s_[2], s_[3]) for j, s_ in enumerate(samples)], out_images=slots)
                free.append(slots)
            pending = [submit(e) for e in plan[:5]]
            nxt = 5
            for _ in range(3):
                finish(pending.pop(0))
                pending.append(submit(plan[nxt])); nxt += 1
            barrier.wait(timeout=180)
            t0 = time.time()
            for _ in range(args.batches):
                finish(pending.pop(0))
                if nxt < len(plan):
                    pending.append(submit(plan[nxt])); nxt += 1
            t1 = time.time()
            out.put((rank, t0, t1, None))
            f.close()
            return
        it = f.epoch(0)
        for _ in range(3):
            next(it)
        torch.cuda.synchronize()
        barrier.wait(timeout=180)
        t0, c0 = time.time(), time.process_time()
        for _ in range(args.batches):
            next(it)
        torch.cuda.synchronize()
        t1, c1 = time.time(), time.process_time()
        out.put((rank, t0, t1, None, c1 - c0))
        it.close()
        f.close()
    except BaseException as e:      # noqa: BLE001 - reported by the parent; never leave the others at the barrier
        barrier.abort()
        out.put((rank, 0.0, 0.0, '%s: %s' % (type(e).__name__, e)))


def concurrent_feeders(args, lines):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    n = args.feeders
    for backend in args.backends.split(','):
        for workers in [int(v) for v in args.workers.split(',')]:
            barrier, out = ctx.Barrier(n), ctx.Queue()
            procs = [ctx.Process(target=_one_feeder, args=(r, n, lines, args, backend, workers, barrier, out)) for r in range(n)]
            for p in procs:
                p.start()
            res = []
            try:
                for _ in range(n):
                    res.append(out.get(timeout=420))
            except Exception:       # noqa: BLE001 - queue.Empty: a feeder never reported
                pass
            for p in procs:
                p.join(timeout=20)
                if p.is_alive():
                    p.terminate()
            bad = [r for r in res if r[3]]
            if len(res) < n or bad:
                print('%d feeders, backend=%s, %d workers: FAILED (%d reports; %s)' % (n, backend, workers, len(res),
                                                                                      '; '.join(str(r[3]) for r in bad)), flush=True)
                continue
            span = max(r[2] for r in res) - min(r[1] for r in res)
            per = [args.batches * args.batch_size / (r[2] - r[1]) for r in res]
            total = n * args.batches * args.batch_size / span
            cpu = [r[4] for r in res if len(r) > 4]
            print('%d feeders at once%s, pixels=%s, backend=%s, %d workers each: aggregate %.0f images/s (per feeder min %.0f / mean %.0f / '
                  'max %.0f) against %d x 697 = %d images/s consumed by %d train steps: %.2fx%s'
                  % (n, ' (host half only)' if args.host_only else '', args.pixels.split(',')[0], backend, workers, total, min(per),
                     sum(per) / n, max(per), n, n * 697, n, total / (n * 697.0),
                     '; %.2f ms of CPU per image' % (1e3 * sum(cpu) / (n * args.batches * args.batch_size)) if cpu else ''), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--host-only', action='store_true', help="with --feeders: the host half only (no upload, no device)")
    ap.add_argument('--feeders', type=int, default=0, help="run this many feeder processes concurrently (one per rank of a node)")
    ap.add_argument('--workers', default='8,16,32')
    ap.add_argument('--backends', default='thread,process')
    ap.add_argument('--native', default='1,0')
    ap.add_argument('--pixels', default='host', help="host,gpu: where the pixel work runs (Feeder(pixels=...); thread backend, "
                                                     "native=1 for gpu); with --feeders the first entry is used")
    ap.add_argument('--batches', type=int, default=12)
    ap.add_argument('--batch_size', type=int, default=64)
    ap.add_argument('--check', action='store_true', help="first compare the first batch of a process-backed feeder with a "
                                                         "thread-backed one (same seed): they must be equal")
    args = ap.parse_args()
    import torch
    from yolov3_tensorflow_amd.feeder import Feeder
    lines = write_set(pathlib.Path(tempfile.mkdtemp()), 256)
    lines = (lines * ((args.batches + 9) * args.batch_size // len(lines) + 1))[:(args.batches + 9) * args.batch_size]
    print('host threads available: %d' % len(os.sched_getaffinity(0)), flush=True)
    if args.feeders > 0:
        concurrent_feeders(args, lines)
        return
    if args.check:
        first = {}
        for backend in ('thread', 'process'):
            f = Feeder(lines[:32], 16, 80, [416, 416], ANCHORS, mode='train', use_mix_up=True, num_threads=4, prefetch=2,
                       seed=5, backend=backend)
            it = f.epoch(0)
            b = next(it)
            torch.cuda.synchronize()
            first[backend] = [b.images.cpu()] + [y.cpu() for y in b.y_true]
            it.close()
            f.close()
        same = all(torch.equal(x, y) for x, y in zip(first['thread'], first['process']))
        print('first batch: process-backed feeder %s thread-backed feeder' % ('==' if same else '!='), flush=True)
    for native in args.native.split(','):
        os.environ['Y3_FEED_NATIVE'] = native
        for backend in args.backends.split(','):
            for workers in [int(v) for v in args.workers.split(',')]:
                for pixels in args.pixels.split(','):
                    if pixels == 'gpu' and (backend != 'thread' or native != '1'):
                        continue
                    f = Feeder(lines, args.batch_size, 80, [416, 416], ANCHORS, mode='train', use_mix_up=True,
                               num_threads=workers, prefetch=5, seed=1, backend=backend, pixels=pixels)
                    it = f.epoch(0)
                    for _ in range(3):              # pool start-up, pinned buffers
                        next(it)
                    torch.cuda.synchronize()
                    t0, c0 = time.perf_counter(), time.process_time()
                    for _ in range(args.batches):
                        next(it)
                    torch.cuda.synchronize()
                    dt, cpu = time.perf_counter() - t0, time.process_time() - c0
                    it.close()
                    f.close()
                    images = args.batches * args.batch_size
                    print('native=%s backend=%-7s workers=%3d pixels=%-4s: %7.0f images/s%s' % (
                        native, backend, workers, pixels, images / dt,
                        ', %.2f ms of this process\'s CPU per image' % (1e3 * cpu / images) if backend == 'thread' else ''), flush=True)


if __name__ == '__main__':
    main()
