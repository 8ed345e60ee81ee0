"""get_kmeans.py (anchor k-means, the tool behind ./data/yolo_anchors.txt) against vectors produced by the REFERENCE's own
get_kmeans.py (tests/golden/make_kmeans_golden.py imports it: pure numpy): same start under the same seed, same Lloyd
iterations, same anchors and mean IoU - bit for bit, although the distances are computed for all boxes at once here.  No GPU."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def test_anchor_kmeans_equals_the_reference(tmp_path):
    import get_kmeans as km
    g = np.load(os.path.join(HERE, 'golden', 'reference_kmeans_goldens.npz'))
    for i in range(4):
        boxes, k = g['boxes_%d' % i], int(g['k_%d' % i])
        np.random.seed(100 + i)                                  # the global generator, as the golden run had it
        anchors, quality = km.get_kmeans(boxes.copy(), k)
        assert anchors == g['anchors_%d' % i].tolist()
        assert quality == float(g['avg_iou_%d' % i])
        # an explicit generator in the same state gives the same result
        again, q2 = km.get_kmeans(boxes.copy(), k, rng=np.random.RandomState(100 + i))
        assert again == anchors and q2 == quality
        np.testing.assert_array_equal(km.iou(boxes[0], boxes[1:k + 1]), g['iou_row_%d' % i])
    np.testing.assert_array_equal(km.translate_boxes(g['translate_in']), g['translate_out'])
    path = str(tmp_path / 'anno.txt')
    with open(path, 'w') as f:
        f.write('\n'.join(str(l) for l in g['anno_lines']) + '\n')
    np.testing.assert_array_equal(km.parse_anno(path, target_size=[416, 416]), g['anno_416'])
    np.testing.assert_array_equal(km.parse_anno(path), g['anno_orig'])


def test_command_line_writes_the_anchor_file_the_loader_reads(tmp_path, capsys):
    import get_kmeans as km
    from yolov3_tensorflow_amd.utils.misc_utils import parse_anchors
    rng = np.random.RandomState(3)
    lines = []
    for i in range(40):
        parts = ['%d' % i, 'img%d.jpg' % i, '640', '480']
        for _ in range(int(rng.randint(1, 6))):
            x0, y0 = rng.uniform(0, 300), rng.uniform(0, 200)
            parts += ['%d' % rng.randint(0, 80), '%.1f' % x0, '%.1f' % y0, '%.1f' % (x0 + rng.uniform(8, 330)),
                      '%.1f' % (y0 + rng.uniform(8, 270))]
        lines.append(' '.join(parts))
    anno, out = str(tmp_path / 'train.txt'), str(tmp_path / 'anchors.txt')
    open(anno, 'w').write('\n'.join(lines) + '\n')
    anchors, quality = km.main([anno, '--target_size', '416', '416', '--cluster_num', '9', '--seed', '4', '--output', out])
    text = capsys.readouterr().out
    assert 'anchors are:' in text and 'the average iou is:' in text
    assert len(anchors) == 9 and 0.3 < quality <= 1.0
    areas = [w * h for w, h in anchors]
    assert areas == sorted(areas)
    np.testing.assert_array_equal(parse_anchors(out), np.asarray(anchors, np.float32))       # [9, 2], the loader's view
    assert km.main([anno, '--seed', '4'])[0] == anchors                                         # seeded: reproducible
    try:
        km.iou([0.0, 5.0], np.array([[3.0, 4.0]]))
    except ValueError as e:
        assert 'no area' in str(e)
    else:
        raise AssertionError('a box without area must be refused, as in the reference')
