"""misc/parse_voc_xml.py and misc/remove_optimizers_params_in_ckpt.py (data / checkpoint preparation around the path).  No GPU."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, 'misc'))

XML = """<annotation><filename>%s.jpg</filename><size><width>%d</width><height>%d</height><depth>3</depth></size>%s</annotation>"""
OBJ = """<object><name>%s</name><difficult>%d</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"""


def test_voc_xml_to_annotation_lines(tmp_path):
    import parse_voc_xml as pv
    from PIL import Image
    from yolov3_tensorflow_amd.utils.data_utils import parse_line
    root = tmp_path / 'VOC2007'
    for d in ('Annotations', 'JPEGImages', 'ImageSets/Main'):
        os.makedirs(root / d)
    open(tmp_path / 'voc.names', 'w').write('aeroplane\nbicycle\ncat\n')
    items = {'000001': (500, 375, [('cat', 0, 10, 20, 200, 300), ('bicycle', 1, 1, 2, 30, 40)]),     # one difficult object
             '000002': (320, 240, [('bicycle', 1, 5, 5, 50, 50)]),                                   # only difficult: skipped
             '000003': (640, 480, [('aeroplane', 0, 100, 50, 600, 400), ('cat', 0, 3, 4, 5, 6)]),
             '000004': (100, 100, [('cat', 0, 1, 1, 9, 9)])}                                         # no image file: skipped
    for stem, (w, h, objs) in items.items():
        open(root / 'Annotations' / (stem + '.xml'), 'w').write(XML % (stem, w, h, ''.join(OBJ % o for o in objs)))
        if stem != '000004':
            Image.new('RGB', (w, h)).save(str(root / 'JPEGImages' / (stem + '.jpg')))
    open(root / 'ImageSets/Main/trainval.txt', 'w').write('000001\n000002\n000003\n000004\n')
    open(root / 'ImageSets/Main/test.txt', 'w').write('000003  1\n')
    done = pv.main(['--names', str(tmp_path / 'voc.names'), '--train', '%s:trainval' % root, '--val', '%s:test' % root,
                    '--train_out', str(tmp_path / 'train.txt'), '--val_out', str(tmp_path / 'val.txt')])
    assert done == {'train': 2, 'val': 1}
    lines = open(tmp_path / 'train.txt').read().splitlines()
    assert lines[0] == '0 %s 500 375 2 10 20 200 300' % (root / 'JPEGImages' / '000001.jpg')
    assert lines[1] == '1 %s 640 480 0 100 50 600 400 2 3 4 5 6' % (root / 'JPEGImages' / '000003.jpg')
    idx, path, boxes, labels, w, h = parse_line(lines[1])                     # the format the whole package reads
    assert (idx, w, h) == (1, 640, 480) and labels.tolist() == [0, 2] and boxes.shape == (2, 4)


def test_checkpoint_shrinks_to_its_variables(tmp_path):
    import remove_optimizers_params_in_ckpt as rm
    rng = np.random.RandomState(0)
    arrays = {'yolov3/darknet53_body/Conv/weights': rng.rand(3, 3, 3, 8).astype(np.float32),
              'yolov3/darknet53_body/Conv/weights/Momentum': rng.rand(3, 3, 3, 8).astype(np.float32),
              'yolov3/darknet53_body/Conv/BatchNorm/gamma': rng.rand(8).astype(np.float32),
              'yolov3/darknet53_body/Conv/BatchNorm/gamma/Adam': rng.rand(8).astype(np.float32),
              'yolov3/darknet53_body/Conv/BatchNorm/gamma/Adam_1': rng.rand(8).astype(np.float32),
              'yolov3/darknet53_body/Conv/BatchNorm/moving_mean': rng.rand(8).astype(np.float32),
              'optimizer/step': np.int64(12), 'global_step': np.float64(12)}
    src = str(tmp_path / 'full.npz')
    np.savez(src, **arrays)
    keep, drop = rm.main([src, '--output', str(tmp_path / 'out' / 'small.npz')])
    assert sorted(drop) == sorted(k for k in arrays if k.endswith(('Momentum', 'Adam', 'Adam_1')) or k in ('optimizer/step', 'global_step'))
    small = np.load(str(tmp_path / 'out' / 'small.npz'))
    assert sorted(small.files) == sorted(keep) and len(keep) == 3
    for k in keep:
        np.testing.assert_array_equal(small[k], arrays[k])
