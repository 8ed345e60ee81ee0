"""CPU tests of the F(4x4,3x3) kernel's tiling (csrc/y3_conv_wino44.hip, w44_tiling + the kernel's two index computations).

The host half is called through the C ABI (y3_conv_stats_blocks(d, 2) = the number of 16-tile blocks the launch uses:
pure host arithmetic, no GPU).  The device half - which input pixel a tile's patch pixel reads, which output pixel a tile's
output pixel writes, image by image or in the MOSAIC of the batch (zero gap rows / columns between the images) - is restated
here formula for formula and run as a direct 3x3 convolution per tile: every output pixel must be written exactly once and
equal the zero-padded convolution of its own image (the reference's conv2d: 'SAME' padding per image, utils/layer_utils.py:9-22).
The GPU parity tests (tests/test_conv_gpu.py, WINO44_CASES) check the kernel itself.
"""
import ctypes

import numpy as np
import pytest


def tiling(n, h, w):
    """w44_tiling: (TH, TW, T, mr, mc); mr = 0: every image tiled on its own."""
    th, tw = (h + 3) // 4, (w + 3) // 4
    best = (th, tw, n * th * tw, 0, 0)
    if h % 4 == 0 and w % 4 == 0:
        return best
    for r in range(1, n + 1):
        if n % r:
            continue
        c = n // r
        th, tw = (r * (h + 1) - 1 + 3) // 4, (c * (w + 1) - 1 + 3) // 4
        if th * tw < best[2]:
            best = (th, tw, th * tw, r, c)
    return best


def conv_by_tiles(x, k):
    """The kernel's index formulas (DMA source offsets, `tinfo` row / column parts), a direct 3x3 per tile in place of the
    Winograd arithmetic.  Returns (y, times each output pixel was written)."""
    n, h, w = x.shape
    TH, TW, T, mr, mc = tiling(n, h, w)
    xf = x.reshape(-1)
    y, cnt = np.zeros(n * h * w), np.zeros(n * h * w, int)
    for t in range(T):
        img0 = 0 if mr else t // (TH * TW)
        r = t - img0 * TH * TW
        ty, tx = r // TW, r % TW
        patch = np.zeros((6, 6))
        for kk in range(6):
            for l in range(6):
                yy, xx, img = 4 * ty - 1 + kk, 4 * tx - 1 + l, img0
                ok = yy >= 0 and xx >= 0
                if mr and ok:
                    ry, cx = yy // (h + 1), xx // (w + 1)
                    yy, xx, img = yy - ry * (h + 1), xx - cx * (w + 1), ry * mc + cx
                    ok = ry < mr and cx < mc
                if ok and yy < h and xx < w:
                    patch[kk, l] = xf[(img * h + yy) * w + xx]
        rpart, cpart = [], []
        for q in range(4):
            yy, xx = 4 * ty + q, 4 * tx + q
            if mr:
                ry, cx = yy // (h + 1), xx // (w + 1)
                yy, xx = yy - ry * (h + 1), xx - cx * (w + 1)
                rpart.append((ry * mc * h + yy) * w if ry < mr and yy < h else -1)
                cpart.append(cx * h * w + xx if cx < mc and xx < w else -1)
            else:
                rpart.append((img0 * h + yy) * w if yy < h else -1)
                cpart.append(xx if xx < w else -1)
        for py in range(4):
            for px in range(4):
                if rpart[py] >= 0 and cpart[px] >= 0:
                    o = rpart[py] + cpart[px]
                    y[o] = (k * patch[py:py + 3, px:px + 3]).sum()
                    cnt[o] += 1
    return y.reshape(n, h, w), cnt


@pytest.mark.parametrize('n,h,w', [(1, 5, 5), (2, 5, 7), (6, 5, 5), (7, 13, 13), (32, 13, 13), (16, 26, 26), (4, 13, 26),
                                   (3, 8, 8), (5, 6, 9), (12, 19, 19), (5, 3, 5), (1, 2, 2)])
def test_every_output_pixel_is_written_once_and_equals_the_per_image_convolution(n, h, w):
    rng = np.random.RandomState(n * 100 + h)
    x, k = rng.standard_normal((n, h, w)), rng.standard_normal((3, 3))
    want = np.zeros((n, h, w))
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for a in range(3):
        for b in range(3):
            want += k[a, b] * xp[:, a:a + h, b:b + w]
    got, cnt = conv_by_tiles(x, k)
    assert (cnt == 1).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_mosaic_is_used_where_it_saves_tiles_and_the_library_agrees():
    from yolov3_tensorflow_amd import build, _lib
    build.build(verbose=False)
    L = _lib.lib()
    blocks = lambda n, h, w, cin=256, cout=512: L.y3_conv_stats_blocks(ctypes.byref(_lib.ConvDesc(n, h, w, cin, 0, cout, 3, 1, 1)), 2)
    # the bench batch: 26-grid 4 x 8 images = 27 x 54 tiles (1,568 image by image), 13-grid 392 (512), 52-grid no mosaic
    assert tiling(32, 26, 26) == (27, 54, 1458, 4, 8) and blocks(32, 26, 26) == 92
    assert tiling(32, 13, 13)[2] == 392 and blocks(32, 13, 13) == 25
    assert tiling(32, 52, 52) == (13, 13, 32 * 169, 0, 0) and blocks(32, 52, 52) == 338
    assert tiling(16, 26, 26)[2:] == (729, 4, 4) and blocks(16, 26, 26) == 46          # the two-stream halves
    assert tiling(64, 26, 26)[2] == 2916 and blocks(64, 26, 26) == 183                 # the train step's batch
    for n, h, w in [(1, 13, 13), (7, 13, 13), (5, 6, 9), (12, 19, 19), (3, 20, 28), (2, 13, 13), (24, 26, 26), (10, 52, 52)]:
        t = tiling(n, h, w)[2]
        assert t <= n * ((h + 3) // 4) * ((w + 3) // 4)
        assert blocks(n, h, w) == (t + 15) // 16, (n, h, w)
    assert blocks(2, 13, 13, cin=48) == 0                                              # not an F(4x4) shape


def test_bf16_tile_choice_at_configs4():
    """The tile the bf16 path picks per layer shape at BASELINE configs[4] (608x608, bs=16, and bs=8 = one of two streams): the
    per-layer measurements of profiles/r05_bf16_tiles.txt as a table, through the library's own dispatch query (host-only ABI call
    y3_conv_bf16_tile; csrc/y3_conv_bf16x.hip choose_tile, csrc/y3_conv_bf16r.hip dispatch_r)."""
    import ctypes
    from yolov3_tensorflow_amd import _lib
    L = _lib.lib()
    tile = lambda n, g, k, s, cin, cout: chr(L.y3_conv_bf16_tile(ctypes.byref(_lib.ConvDesc(n, g, g, cin, 0, cout, k, s, 1))))
    want16 = {   # (grid of the INPUT, k, stride, cin, cout): tile at bs=16
        (608, 3, 1, 3, 32): 's', (608, 3, 2, 32, 64): 'x', (304, 1, 1, 64, 32): 'o', (304, 3, 1, 32, 64): 'x',
        (304, 3, 2, 64, 128): 'E', (152, 1, 1, 128, 64): 'o', (152, 3, 1, 64, 128): 'E', (152, 3, 2, 128, 256): 'E',
        (76, 1, 1, 256, 128): 'o', (76, 3, 1, 128, 256): 'E', (76, 3, 2, 256, 512): 'D', (38, 1, 1, 512, 256): 'd',
        (38, 3, 1, 256, 512): 'D', (38, 3, 2, 512, 1024): 'E', (19, 1, 1, 1024, 512): 'e', (19, 3, 1, 512, 1024): 'E',
        (19, 1, 1, 1024, 255): 'f', (38, 1, 1, 768, 256): 'd', (38, 1, 1, 512, 255): 'd', (76, 1, 1, 384, 128): 'o',
        (76, 1, 1, 256, 255): 'o',
    }
    for (g, k, s, cin, cout), t in want16.items():
        assert tile(16, g, k, s, cin, cout) == t, (g, k, s, cin, cout, tile(16, g, k, s, cin, cout), t)
    # half the batch (one of two streams): the deep layers move to the tiles that still give ~a tile per CU
    assert tile(8, 38, 3, 1, 256, 512) == 'E' and tile(8, 19, 3, 1, 512, 1024) == 'E' and tile(8, 38, 1, 1, 512, 256) == 'e'
    assert tile(8, 19, 1, 1, 1024, 512) == 'f'
