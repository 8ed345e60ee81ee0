"""Golden vectors of the reference's augmentation box arithmetic, produced by IMPORTING /root/reference/utils/data_aug.py
unmodified (under a stub `cv2` whose `flip` is a numpy flip - the only cv2 call on the paths exercised here):

    python tests/golden/make_aug_golden.py          # writes tests/golden/reference_aug_goldens.npz (committed)

Every case seeds the process-global `random` / `numpy.random` generators the reference draws from; the product's functions
(yolov3_tensorflow_amd/utils/data_aug.py), handed the SAME global generators under the SAME seeds, must reproduce the
boxes, crops, offsets and mix-up weights exactly (tests/test_feeder_cpu.py).  Pinned: bbox_iou, bbox_crop,
random_crop_with_constraints, random_expand, random_flip, mix_up, resize_with_bbox's box map, and the multi-scale size
sequence of get_batch_data (utils/data_utils.py:193-197).  Not pinned: pixel values that go through cv2 (colour jitter,
resize interpolation) - no OpenCV here.
"""
import os
import random
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    cv2 = types.ModuleType('cv2')
    cv2.flip = lambda img, code: img[:, ::-1] if code == 1 else img[::-1]
    sys.modules['cv2'] = cv2
    sys.path.insert(0, REF)
    from utils import data_aug        # noqa: the reference's own module
    return data_aug


def boxes5(rng, n, w, h):
    x0, y0 = rng.uniform(0, w * 0.7, n), rng.uniform(0, h * 0.7, n)
    x1, y1 = x0 + rng.uniform(8, w * 0.3, n), y0 + rng.uniform(8, h * 0.3, n)
    return np.stack([x0, y0, x1, y1, rng.uniform(0.2, 1.0, n)], 1).astype(np.float32)


def main():
    da = import_reference()
    out = {}
    rng = np.random.RandomState(11)
    a, b = boxes5(rng, 9, 640, 480)[:, :4], boxes5(rng, 5, 640, 480)[:, :4]
    out['iou_a'], out['iou_b'], out['iou_out'] = a, b, da.bbox_iou(a, b)
    bb = boxes5(rng, 12, 640, 480)
    out['crop_in'] = bb
    for i, (crop, aoc) in enumerate((((100, 60, 300, 260), True), ((100, 60, 300, 260), False), ((0, 0, 640, 480), False))):
        out['crop_box_%d' % i] = np.array(crop, np.int64)
        out['crop_aoc_%d' % i] = np.bool_(aoc)
        out['crop_out_%d' % i] = da.bbox_crop(bb, crop, allow_outside_center=aoc)
    # random_crop_with_constraints: global generators seeded per case
    cases = []
    for seed in range(12):
        n = 1 + seed % 5
        box = boxes5(np.random.RandomState(100 + seed), n, 500, 375)
        random.seed(seed)
        np.random.seed(seed)
        nb, crop = da.random_crop_with_constraints(box.copy(), (500, 375))
        out['rc_in_%d' % seed], out['rc_out_%d' % seed], out['rc_crop_%d' % seed] = box, nb, np.array(crop, np.int64)
        cases.append(seed)
    out['rc_seeds'] = np.array(cases, np.int64)
    # random_expand / random_flip / mix_up on small images
    for seed in range(6):
        img = np.random.RandomState(200 + seed).randint(0, 256, (37, 53, 3)).astype(np.uint8)
        box = boxes5(np.random.RandomState(300 + seed), 3, 53, 37)
        random.seed(seed)
        np.random.seed(seed)
        e_img, e_box = da.random_expand(img.copy(), box.copy(), 4)
        out['ex_img_%d' % seed], out['ex_in_%d' % seed] = img, box
        out['ex_shape_%d' % seed], out['ex_out_%d' % seed] = np.array(e_img.shape, np.int64), e_box
        out['ex_sum_%d' % seed] = np.int64(e_img.astype(np.int64).sum())
        random.seed(seed)
        np.random.seed(seed)
        f_img, f_box = da.random_flip(img.copy(), box.copy(), px=0.5, py=0.3)
        out['fl_out_%d' % seed], out['fl_img_%d' % seed] = f_box, np.ascontiguousarray(f_img)
        img2 = np.random.RandomState(400 + seed).randint(0, 256, (41, 47, 3)).astype(np.uint8)
        box2 = boxes5(np.random.RandomState(500 + seed), 2, 47, 41)[:, :4]
        np.random.seed(seed)
        m_img, m_box = da.mix_up(img, img2, box[:, :4], box2)
        out['mx_img2_%d' % seed], out['mx_in2_%d' % seed] = img2, box2
        out['mx_img_%d' % seed], out['mx_out_%d' % seed] = m_img, m_box
    # the multi-scale size sequence of get_batch_data (utils/data_utils.py:193-197), restated call for call
    sizes = []
    for cnt in range(120):
        random.seed(cnt // 10)
        sizes.append(random.sample([[x * 32, x * 32] for x in range(10, 20)], 1)[0])
    out['ms_sizes'] = np.array(sizes, np.int64)
    np.savez_compressed(os.path.join(OUT, 'reference_aug_goldens.npz'), **out)
    print('wrote', os.path.join(OUT, 'reference_aug_goldens.npz'), len(out), 'arrays')


if __name__ == '__main__':
    main()
