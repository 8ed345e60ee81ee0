"""Golden vectors of the reference's anchor k-means, produced by IMPORTING /root/reference/get_kmeans.py (pure numpy) in the
build container:

    python tests/golden/make_kmeans_golden.py        ->  tests/golden/reference_kmeans_goldens.npz

The reference reseeds numpy from the OS inside kmeans(); for a reproducible vector that one call is neutralised here
(np.random.seed -> no-op while kmeans runs) and the global generator is seeded before each case instead.  The twin at the
repository root draws its start from the same global generator when given none, so the same seed gives the same start.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def cases():
    rng = np.random.RandomState(20)
    out = []
    for n, k in ((60, 3), (400, 9), (1500, 9), (97, 5)):
        wh = np.exp(rng.normal(3.5, 0.9, (n, 2))) + 1.0           # box sizes from a few to a few hundred pixels
        out.append((np.round(wh, 1), k))
    return out


def main():
    sys.path.insert(0, REF)
    import get_kmeans as ref
    real_seed = np.random.seed
    res = {}
    for i, (boxes, k) in enumerate(cases()):
        np.random.seed(100 + i)
        np.random.seed = lambda *a, **kw: None
        try:
            anchors, quality = ref.get_kmeans(boxes.copy(), k)
        finally:
            np.random.seed = real_seed
        res['boxes_%d' % i], res['k_%d' % i] = boxes, np.int64(k)
        res['anchors_%d' % i], res['avg_iou_%d' % i] = np.asarray(anchors, np.int64), np.float64(quality)
        res['iou_row_%d' % i] = ref.iou(boxes[0], boxes[1:k + 1])
    corners = np.array([[10., 20., 50., 90.], [30., 5., 12., 40.], [0., 0., 7., 3.]])
    res['translate_in'], res['translate_out'] = corners, ref.translate_boxes(corners)
    # an annotation file in the reference's line format, parsed on both scales
    lines = ['0 a.jpg 640 480 3 10.5 20 110.5 220 7 300 100 420 400.25', '1 b.jpg 500 375 1 1 2 3.5 9', '2 c.jpg 200 800 0 50 60 150 700']
    path = os.path.join(HERE, '_kmeans_anno.txt')
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    res['anno_lines'] = np.array(lines)
    res['anno_416'] = ref.parse_anno(path, target_size=[416, 416])
    res['anno_orig'] = ref.parse_anno(path)
    os.remove(path)
    np.savez_compressed(os.path.join(HERE, 'reference_kmeans_goldens.npz'), **res)
    print('wrote reference_kmeans_goldens.npz:', {k: v.shape for k, v in res.items() if k.startswith('anchors')})


if __name__ == '__main__':
    main()
