"""Data-parallel gradient exchange for the train step (the reference itself is single-GPU,
README.md:140,210, so this is north-star-defined): every rank owns ONE flat fp32 gradient buffer laid out in the order
the backward pass PRODUCES the gradients (last layer first).  The buffer is cut into buckets of `bucket_bytes`; as soon
as backward has written everything below a bucket's upper edge, that bucket's all-reduce (SUM) is issued asynchronously
— on RCCL it runs on the process group's own stream, behind an event on the compute stream, so it overlaps the rest of
backward — and `finish()` joins them before the clip/update kernels read the buffer.  The 1/world average is not a
separate pass: it is folded into the clip/update kernel (`grad_scale`).

xGMI is point-to-point (7 links x ~153 GB/s per GPU) and a ring all-reduce is bound by one link, so the bucket is large
(32 MiB default: 8 buckets for the 247.8 MB whole-model gradient, 3 for the 85.5 MB head) — enough to hide the ~20 us
launch/sync latency of a collective, small enough that the last bucket (the first layers' gradients, produced last)
is a small tail.

The class is tensor-agnostic (CPU tensors on `gloo`, device tensors on `nccl` = RCCL), which is how
tests/test_distributed_cpu.py drives this exact code path with world_size 2 on CPU.
"""
import torch
import torch.distributed as dist

DEFAULT_BUCKET_BYTES = 32 << 20


def plan_buckets(segment_ends, bucket_bytes=DEFAULT_BUCKET_BYTES, itemsize=4):
    """Cut [0, segment_ends[-1]) into buckets whose edges are segment edges (a segment = one layer's gradients,
    in production order) and whose size is the smallest run of whole segments reaching `bucket_bytes`.
    Returns the list of bucket upper edges (element offsets, strictly increasing, last == segment_ends[-1])."""
    edges, start = [], 0
    target = max(1, int(bucket_bytes) // itemsize)
    for e in segment_ends:
        if e - start >= target:
            edges.append(int(e))
            start = e
    total = int(segment_ends[-1]) if len(segment_ends) else 0
    if total > start:
        edges.append(total)
    return edges


class GradientExchange(object):
    """Bucketed, overlapped all-reduce of one flat gradient buffer.

        ex = GradientExchange(flat, segment_ends, group)
        ... backward writes flat[0:e0], then flat[e0:e1], ...; after each layer:   ex.ready(upto=e_k)
        ex.finish()          # all buckets reduced (SUM over ranks) and visible to the current stream
        scale = ex.grad_scale    # 1/world, applied by the consumer

    With world == 1 (or no initialised process group) every call is a no-op."""

    def __init__(self, flat, segment_ends, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
        self.flat = flat
        self.group = group
        self.world = 1
        if group is not None or dist.is_initialized():
            self.world = dist.get_world_size(group)
        self.edges = plan_buckets(list(segment_ends), bucket_bytes, flat.element_size())
        self.grad_scale = 1.0 / self.world
        # test hooks (tests/test_rccl_gpu.py, the only RCCL evidence a one-GPU box can give): `force` issues the
        # collectives even in a one-rank group, `op` replaces SUM (a pre-multiplied sum makes a one-rank all-reduce
        # change the values, so a stream-ordering mistake shows up in the numbers)
        self.force = False
        self.op = dist.ReduceOp.SUM
        self._next = 0          # next bucket to issue
        self._works = []
        self.issued = []        # (lo, hi) of every bucket issued in this step, in order (introspection / tests)

    def begin(self):
        self._next = 0
        self._works = []
        self.issued = []

    def _issue(self, lo, hi):
        self.issued.append((lo, hi))
        if (self.world > 1 or self.force) and hi > lo:
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=self.op, group=self.group, async_op=True))

    def ready(self, upto):
        """Everything in flat[0:upto) has been written (enqueued on the current stream)."""
        while self._next < len(self.edges) and self.edges[self._next] <= upto:
            lo = self.edges[self._next - 1] if self._next else 0
            self._issue(lo, self.edges[self._next])
            self._next += 1

    def finish(self):
        """Issue what is left and make every reduced bucket visible to the current stream."""
        self.ready(self.edges[-1] if self.edges else 0)
        for w in self._works:
            w.wait()
        self._works = []


def all_reduce_mean_(flat, segment_ends=None, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
    """One-shot form (no overlap): SUM-all-reduce `flat` bucket by bucket and scale by 1/world in place."""
    ends = list(segment_ends) if segment_ends is not None else [flat.numel()]
    ex = GradientExchange(flat, ends, group, bucket_bytes)
    ex.begin()
    ex.finish()
    if ex.world > 1:
        flat.mul_(ex.grad_scale)
    return flat
