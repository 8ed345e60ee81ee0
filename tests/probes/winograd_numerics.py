"""Numerics probe (not a test, not product code): what would F(4x4,3x3) Winograd cost in accuracy on THIS network?

The CPU oracle's forward (oracle/yolo_ref.py) is run with its stride-1 3x3 convs (Cin >= 32: the layers the GPU's Winograd
kernel takes) replaced by an fp32 Winograd restatement - F(2x2,3x3) as the product uses, and F(4x4,3x3) with the standard
interpolation points (0, +-1, +-2, inf) - and compared with the fp64 direct forward: feature maps, and decoded boxes
relative to the box scale (the 1e-3 of BASELINE.json's north star).

    python tests/probes/winograd_numerics.py            (CPU only, ~2 minutes)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                  [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
}


def winograd_conv(x, w_oihw, m):
    """Stride-1 SAME 3x3 conv of NCHW x as F(m x m, 3x3), every step in x.dtype (fp32: products and sums like the GPU's)."""
    bt, g, at = (torch.from_numpy(a).to(x.dtype) for a in MATS[m])
    n, c, h, w = x.shape
    o = w_oihw.shape[0]
    th, tw = -(-h // m), -(-w // m)
    xp = F.pad(x, (1, tw * m - w + 1, 1, th * m - h + 1))
    d = xp.unfold(2, m + 2, m).unfold(3, m + 2, m)                       # [n, c, th, tw, a, a]
    v = torch.einsum('ia,nctuab,jb->ijntuc', bt, d, bt)                  # B^T d B
    u = torch.einsum('ia,ocab,jb->ijco', g, w_oihw, g)                   # G g G^T
    a = m + 2
    mm = torch.matmul(v.reshape(a, a, n * th * tw, c), u)                # [a, a, n*th*tw, o]
    y = torch.einsum('pi,ijtq,rj->tqpr', at, mm, at)                     # A^T M A -> [n*th*tw, o, m, m]
    y = y.reshape(n, th, tw, o, m, m).permute(0, 3, 1, 4, 2, 5).reshape(n, o, th * m, tw * m)
    return y[:, :, :h, :w].contiguous()


def main():
    from conftest import blob_images, COCO_ANCHORS
    from oracle import yolo_ref
    params = yolo_ref.synthetic_params(80, seed=1)
    x = blob_images(0, 2, 416)
    real_conv = F.conv2d
    ref = yolo_ref.forward(params, x, dtype=torch.float64)
    boxes_ref, confs_ref, probs_ref = yolo_ref.predict(ref, COCO_ANCHORS, [416, 416], 80, dtype=np.float64)
    scale = np.maximum(np.abs(boxes_ref).max(axis=-1, keepdims=True), 1.0)
    print('variant                 feature maps max|d|   boxes max rel to box scale   confs max|d|   probs max|d|')
    for name, m in (('direct fp32', 0), ('F(2x2,3x3) fp32', 2), ('F(4x4,3x3) fp32', 4)):
        def conv(inp, wt, bias=None, stride=1, padding=0, _m=m):
            if _m and wt.shape[2] == 3 and stride == 1 and wt.shape[1] >= 32:
                return winograd_conv(inp, wt, _m)
            return real_conv(inp, wt, bias, stride=stride, padding=padding)
        yolo_ref.F.conv2d = conv
        try:
            fms = yolo_ref.forward(params, x, dtype=torch.float32)
        finally:
            yolo_ref.F.conv2d = real_conv
        b, c, p = yolo_ref.predict(fms, COCO_ANCHORS, [416, 416], 80, dtype=np.float32)
        dfm = max(float(np.abs(f.astype(np.float64) - r).max()) for f, r in zip(fms, ref))
        print('%-22s  %.3e             %.3e                    %.3e      %.3e'
              % (name, dfm, float((np.abs(b - boxes_ref) / scale).max()), float(np.abs(c - confs_ref).max()),
                 float(np.abs(p - probs_ref).max())), flush=True)


if __name__ == '__main__':
    main()
