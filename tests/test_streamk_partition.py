# coding: utf-8
"""The stream-K work split and its in-kernel hand-off protocol, replayed on the host (no GPU).

The conv kernels (y3_conv_common.h sk_range / y3_conv_wino.hip wk_range) divide (unit, K-step) items among
persistent workgroups (the Winograd kernel: whole rounds of blocks first, kind 2) and finish a cut unit inside the kernel: the worker owning the unit's K-step 0 (its LAST
segment) adds the partial sums published by the following local workers of its group (their FIRST segments).
`y3_streamk_range` evaluates the same functions on the host; this test replays the protocol on the ranges it
returns for the network's layer shapes and checks the invariants the device code relies on:
  * every group's ranges tile its whole units exactly, in order, with no gaps;
  * a worker publishes at most once, and only from its first segment;
  * the partials a consumer counts cover exactly the rest of its unit, in K order;
  * every producer a consumer waits for runs in a workgroup with a smaller id (dispatched earlier): no deadlock.
"""
import ctypes

import pytest

from yolov3_tensorflow_amd import _lib


def ranges(kind, units, ksteps, workers):
    L = _lib.lib()
    G = workers // 8
    out = {}
    for x in range(8):
        for j in range(G):
            b, e = ctypes.c_longlong(), ctypes.c_longlong()
            _lib.check(L.y3_streamk_range(kind, units, ksteps, workers, x, j, ctypes.byref(b), ctypes.byref(e)))
            out[(x, j)] = (b.value, e.value)
    return out


def group_blocks(units, x):
    q, r = divmod(units, 8)
    b0 = x * q + min(x, r)
    return b0, b0 + q + (1 if x < r else 0)


def replay(kind, units, ksteps, workers):
    G = workers // 8
    rng = ranges(kind, units, ksteps, workers)
    covered = 0
    published = {}           # worker -> (unit, first K-step, end K-step)
    finished = {}            # unit -> K-steps accounted for by its finishing worker
    if kind == 2:
        # hybrid: whole rounds of every group's blocks are computed uncut (round r, local worker j: block b0 + r*G + j);
        # the ranges returned describe the remaining blocks only
        for x in range(8):
            b0, b1 = group_blocks(units, x)
            R = (b1 - b0) // G
            for r in range(R):
                for j in range(G):
                    unit = b0 + r * G + j
                    assert unit not in finished
                    finished[unit] = ksteps
            covered += R * G * ksteps
            assert rng[(x, 0)][0] == (b0 + R * G) * ksteps and rng[(x, G - 1)][1] == b1 * ksteps
            assert b1 - (b0 + R * G) < G                        # less than one round is ever cut
    for x in range(8):
        pos = None
        for j in range(G):
            b, e = rng[(x, j)]
            assert b <= e
            if pos is None:
                assert b % ksteps == 0                      # a group starts on a unit boundary
            else:
                assert b == pos                             # contiguous, in local-worker order
            pos = e
            covered += e - b
        assert pos % ksteps == 0                            # ... and ends on one
    assert covered == units * ksteps
    for (x, j), (b, e) in rng.items():
        item, first = b, True
        while item < e:
            unit, ks = divmod(item, ksteps)
            unit_end = (unit + 1) * ksteps
            seg_end = min(unit_end, e)
            if ks > 0:                                      # producer: only ever the first segment
                assert first and (x, j) not in published
                published[(x, j)] = (unit, ks, seg_end - unit * ksteps)
            else:
                have = seg_end - unit * ksteps
                if seg_end < unit_end:                      # consumer: must be the last segment of the range
                    assert seg_end == e
                    jj = j + 1
                    while jj < G and rng[(x, jj)][0] < unit_end:
                        pb, pe = rng[(x, jj)]
                        if pb < pe:
                            # workgroup id = x + 8 * (G - 1 - local worker): the producer's is smaller
                            assert x + 8 * (G - 1 - jj) < x + 8 * (G - 1 - j)
                            assert pb == unit * ksteps + have, 'partials must continue the K range in order'
                            have = min(pe, unit_end) - unit * ksteps
                        jj += 1
                assert have == ksteps, (kind, units, ksteps, unit, have)
                assert unit not in finished
                finished[unit] = have
            item, first = seg_end, False
    assert len(finished) == units
    # every published partial is consumed by exactly the unit it belongs to (checked above through `have`)
    for (x, j), (unit, k0, k1) in published.items():
        assert 0 < k0 < k1 <= ksteps


# (units, K-steps) of the stream-K launches of the 416x416 bs=32 forward and of a few awkward sizes
DIRECT = [(1352, 36), (680, 72), (344, 144), (344, 288), (56, 72), (33, 18), (32, 16), (2047, 9), (100, 5), (1000, 1)]
WINO = [(1352, 16), (680, 32), (400, 64), (304, 8), (257, 4), (2047, 128), (511, 3)]


@pytest.mark.parametrize('units,ksteps', DIRECT)
def test_direct_streamk_partition(units, ksteps):
    replay(0, units, ksteps, 512)


@pytest.mark.parametrize('units,ksteps', WINO)
def test_winograd_streamk_partition(units, ksteps):
    replay(1, units, ksteps, 256)


@pytest.mark.parametrize('units,ksteps', WINO + [(256, 16), (264, 4), (3000, 2)])
def test_winograd_hybrid_streamk_partition(units, ksteps):
    replay(2, units, ksteps, 256)


def test_streamk_range_rejects_bad_arguments():
    L = _lib.lib()
    b, e = ctypes.c_longlong(), ctypes.c_longlong()
    for args in ((0, 0, 4, 512, 0, 0), (0, 10, 4, 500, 0, 0), (0, 10, 4, 512, 8, 0), (0, 10, 4, 512, 0, 64),
                 (0, 1 << 20, 1 << 12, 512, 0, 0)):
        assert L.y3_streamk_range(*args, ctypes.byref(b), ctypes.byref(e)) != 0
