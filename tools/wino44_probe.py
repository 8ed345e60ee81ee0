#!/usr/bin/env python
"""Clock probe of conv_wino44_f32_kernel (a -DW44_PROBE variant build, tools/build_variant.py): wave 0 of every workgroup
stamps s_memtime at its phase boundaries; prints the mean cycles per phase over the workgroups of one launch, per shape.

    Y3_LIB_PATH=tools/_probe/lib_probe.so python tools/wino44_probe.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAMES = ['setup', 'dma+wait', 'transform0', 'kstep0', 'kstep1', 'ksteps mid', 'kstep last', 'drain+AtMA+stage', 'barrier',
         'stores', 'barrier2']


def main():
    from yolov3_tensorflow_amd import engine, _lib
    lib = _lib.lib()
    fn = lib.y3_debug_w44_probe
    fn.argtypes = [ctypes.c_void_p]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    bt = int(os.environ.get('W44_BT', '16'))
    shapes = ((104, 64, 128), (52, 128, 256), (26, 256, 512), (13, 512, 1024))
    if len(sys.argv) > 2:
        shapes = ((int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])),)
    for g, cin, cout in shapes:
        x = torch.rand((n, g, g, cin), device='cuda')
        w = torch.randn((3, 3, cin, cout), device='cuda') * 0.05
        sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
        res = torch.rand((n, g, g, cout), device='cuda')
        w4 = engine.pack_wino44(w)
        tiles = n * ((g + 3) // 4) ** 2
        blocks = ((tiles + bt - 1) // bt) * (cout // 64)
        buf = torch.zeros((blocks, 32), dtype=torch.int64, device='cuda')
        for _ in range(3):
            engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, residual=res, use_workspace=False)
        torch.cuda.synchronize()
        assert fn(buf.data_ptr()) == 0
        engine.conv2d_fwd_wino44(x, w4, sc, sh, cout, True, residual=res, use_workspace=False)
        torch.cuda.synchronize()
        fn(None)
        t = buf.cpu().double()
        t = t[t[:, 0] > 0]
        start = t[:, 0] - t[:, 0].min()
        d = (t[:, 1:12] - t[:, 0:11])
        total = t[:, 11] - t[:, 0]
        ks = cin // 8
        print('%dx%d %d->%d: %d blocks, %d K-steps; block start (cycles after the first) p50 %.0f max %.0f; block total mean %.0f'
              % (g, g, cin, cout, blocks, ks, start.median(), start.max(), total.mean()))
        m = d.mean(0)
        for i, nm in enumerate(NAMES[:11]):
            extra = ''
            if nm == 'ksteps mid' and ks > 3:
                extra = '  (%.0f per K-step over %d)' % (m[i] / (ks - 3), ks - 3)
            print('    %-12s %9.0f%s' % (nm, m[i], extra))
        k = (t[:, 17:24] - t[:, 16:23]).mean(0)
        print('    K-step ks0+3 of wave 0: dma issue + transform %.0f | frag preload + pair 0 %.0f | pairs 1-5 %.0f | pairs 6-11 %.0f | pairs 12-17 %.0f | waitcnt %.0f | barrier %.0f | sum %.0f'
              % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], k.sum()))
        span = (t[:, 11].max() - t[:, 0].min())
        print('    launch span %.0f cycles (s_memtime ticks at 100 MHz if constant: see README)' % span)


if __name__ == '__main__':
    main()
