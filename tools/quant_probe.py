"""Probe: same conv (3x3 128->256) with M chosen so the block count is / is not a multiple of the 512
co-resident slots -> separates tile-quantisation loss from steady-state efficiency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov3_tensorflow_amd import engine, framework as fw, _lib
dev = fw.default_device(); L = _lib.lib()
k, cin, cout = 3, 128, 256
w = torch.randn((k, k, cin, cout), device=dev) * 0.03
wp = torch.empty(k * k * cout * cin, device=dev)
_lib.check(L.y3_pack_conv_weights(fw.context(), fw.ptr(w), k, cin, cout, fw.ptr(wp)))
sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
for (n, h) in ((16, 64), (24, 64), (32, 64), (64, 64), (32, 52), (8, 64), (4, 64)):
    x = torch.randn((n, h, h, cin), device=dev)
    for _ in range(3): engine.conv2d_fwd(x, wp, sc, sh, k, 1, cout, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): engine.conv2d_fwd(x, wp, sc, sh, k, 1, cout, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    M = n * h * h
    blocks = ((M + 127) // 128) * 2
    print('N=%d H=%d M=%d blocks=%d (%.2f x 512): %.4f ms %.1f TF/s' % (n, h, M, blocks, blocks / 512.0, ms, 2.0 * 9 * cin * cout * M / ms / 1e9))
