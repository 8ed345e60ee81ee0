# coding: utf-8
"""The feeder's pixel work on the device: a batch of feed_native.PixelJob -> the float32 image batch in HBM.

Host: y3f_plan_batch (liby3feed.so) writes one blob per batch into a recycled pinned buffer - the job records, the source
pixels the crop windows can see, the jitter maps, Pillow's filter tables -; one asynchronous H2D copy (8-bit sources: a
third of the bytes of the float32 batch the host path uploads); then y3_feed_run (libyolo355.so): three launches.  The
result is bit-identical to feed_native.sample (tests/test_feed_gpu.py).  There is no CPU form of this module: without a
HIP device it raises.
"""
import ctypes

import numpy as np

from . import _lib, feed_native
from . import framework as fw


class DevicePixels(object):
    """Per device: the conversion tables (uploaded once) and a pool of pinned blob buffers recycled once their upload
    has completed.  run() enqueues everything on torch's CURRENT stream of the device."""

    def __init__(self, device=None):
        import torch
        self.device = torch.device(device) if device is not None else fw.default_device()
        self.tables = torch.from_numpy(feed_native.device_tables()).to(self.device)
        self.busy = []          # (event, pinned uint8 tensor)

    def _take(self, nbytes):
        import torch
        free = [i for i, (ev, _) in enumerate(self.busy) if ev.query()]
        best = None
        for i in free:
            if self.busy[i][1].numel() >= nbytes and (best is None or self.busy[i][1].numel() < self.busy[best][1].numel()):
                best = i
        if best is not None:
            return self.busy.pop(best)[1]
        for i in reversed(free[:-2]):           # too small for this batch: keep a couple, unpin the rest
            del self.busy[i]
        return torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()

    def run(self, pixel_jobs, threads=0):
        """[PixelJob] (all with the same output size) -> float32 device tensor [n, out_h, out_w, 3]."""
        import torch
        n = len(pixel_jobs)
        if n == 0:
            raise ValueError("DevicePixels.run: no jobs")
        jobs = feed_native.job_array(pixel_jobs)
        need, scratch_bytes = feed_native.plan_sizes(jobs, n)
        pinned = self._take(need)
        got, scratch_bytes = feed_native.plan_into(jobs, n, pinned.data_ptr(), pinned.numel(), threads)
        assert got == need
        out_h, out_w = pixel_jobs[0].job.out_h, pixel_jobs[0].job.out_w
        with torch.cuda.device(self.device):
            blob = torch.empty(need, dtype=torch.uint8, device=self.device)
            blob.copy_(pinned[:need], non_blocking=True)
            scratch = torch.empty(max(scratch_bytes, 16), dtype=torch.uint8, device=self.device)
            out = torch.empty((n, out_h, out_w, 3), dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().y3_feed_run(fw.context(self.device), ctypes.c_void_p(blob.data_ptr()),
                                              ctypes.c_void_p(pinned.data_ptr()), n, ctypes.c_void_p(self.tables.data_ptr()),
                                              ctypes.c_void_p(scratch.data_ptr()), scratch.numel(),
                                              ctypes.c_void_p(out.data_ptr()), out_h, out_w))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        self.busy.append((ev, pinned))
        # blob and scratch go back to torch's allocator here: it hands them out again only behind this stream's work
        return out


def sample_batch(cases, device=None):
    """feed_native.sample for a list of argument dicts, on the device (tests, tools)."""
    return DevicePixels(device).run([feed_native.make_job(**c) for c in cases])
