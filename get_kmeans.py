# coding: utf-8
"""Anchor boxes for a dataset: k-means over the (width, height) of its ground-truth boxes under the 1 - IoU distance, the
tool that produces ./data/yolo_anchors.txt for the reference (its get_kmeans.py) - same functions, same results.

What differs from the reference's script: the distances of ALL boxes to the k centres are one numpy expression per Lloyd
iteration instead of a Python loop over boxes (the same element-wise arithmetic, so the same numbers; COCO's 860,000 boxes
take seconds instead of hours), the random start comes from a generator the caller can seed (the reference reseeds from
the OS inside kmeans()), and the command line takes the annotation file, the target size and k instead of constants:

    python get_kmeans.py ./data/my_data/train.txt --target_size 416 416 --cluster_num 9 [--output ./data/yolo_anchors.txt]
"""
from __future__ import division, print_function

import argparse
import sys

import numpy as np


def iou(box, clusters):
    """IoU of one (width, height) box with k (width, height) centres, all anchored at the origin -> [k]."""
    return _iou_matrix(np.asarray(box, np.float64).reshape(1, 2), clusters)[0]


def _iou_matrix(boxes, clusters):
    """[r, 2] x [k, 2] -> [r, k]; the arithmetic of the reference's iou() (get_kmeans.py:8-29) for every pair."""
    w = np.minimum(clusters[None, :, 0], boxes[:, None, 0])
    h = np.minimum(clusters[None, :, 1], boxes[:, None, 1])
    if np.count_nonzero(w == 0) > 0 or np.count_nonzero(h == 0) > 0:
        raise ValueError("Box has no area")
    overlap = w * h
    area_b = boxes[:, 0] * boxes[:, 1]
    area_c = clusters[:, 0] * clusters[:, 1]
    return np.true_divide(overlap, area_b[:, None] + area_c[None, :] - overlap + 1e-10)


def avg_iou(boxes, clusters):
    """Mean over the boxes of the IoU with their best centre (get_kmeans.py:32-41)."""
    return np.mean(np.max(_iou_matrix(boxes, clusters), axis=1))


def translate_boxes(boxes):
    """[r, 4] corner boxes -> [r, 2] (|x_max - x_min|, |y_max - y_min|) (get_kmeans.py:44-56)."""
    boxes = np.asarray(boxes)
    return np.abs(boxes[:, 2:4] - boxes[:, 0:2])


def kmeans(boxes, k, dist=np.median, rng=None):
    """Lloyd iterations with the 1 - IoU distance from k distinct boxes drawn at random (Forgy start) until no box changes
    its centre; a centre moves to `dist` (median) of its boxes (get_kmeans.py:59-93).  rng: numpy generator for the start
    (default: the global one, NOT reseeded here)."""
    rng = np.random if rng is None else rng
    rows = boxes.shape[0]
    clusters = boxes[rng.choice(rows, k, replace=False)]
    assigned = np.zeros((rows,))
    while True:
        nearest = np.argmin(1 - _iou_matrix(boxes, clusters), axis=1)
        if (assigned == nearest).all():
            return clusters
        for c in range(k):
            clusters[c] = dist(boxes[nearest == c], axis=0)
        assigned = nearest


def parse_anno(annotation_path, target_size=None):
    """(width, height) of every box of an annotation file (`idx path img_w img_h [label x0 y0 x1 y1]...` per line), on the
    letterboxed target_size = [width, height] scale when given, else in original pixels (get_kmeans.py:96-122)."""
    sizes = []
    with open(annotation_path, 'r') as f:
        for line in f:
            fields = line.strip().split(' ')
            if len(fields) < 4:
                continue
            img_w, img_h = int(fields[2]), int(fields[3])
            scale = 1.0 if target_size is None else min(target_size[0] / img_w, target_size[1] / img_h)
            boxes = fields[4:]
            for i in range(len(boxes) // 5):
                x0, y0, x1, y1 = (float(v) for v in boxes[5 * i + 1:5 * i + 5])
                assert x1 - x0 > 0 and y1 - y0 > 0, 'box without area in: %s' % line.strip()
                if target_size is None:
                    sizes.append([x1 - x0, y1 - y0])
                else:
                    width, height = x1 - x0, y1 - y0
                    width *= scale
                    height *= scale
                    sizes.append([width, height])
    return np.asarray(sizes)


def get_kmeans(anno, cluster_num=9, rng=None):
    """(anchors as integer [w, h] pairs sorted by area, mean best IoU) (get_kmeans.py:125-135)."""
    centres = kmeans(anno, cluster_num, rng=rng)
    quality = avg_iou(anno, centres)
    return sorted(centres.astype('int').tolist(), key=lambda wh: wh[0] * wh[1]), quality


def main(argv=None):
    ap = argparse.ArgumentParser(description="k-means anchors for an annotation file")
    ap.add_argument('annotation_path', nargs='?', default='train.txt')
    ap.add_argument('--target_size', nargs=2, type=int, default=[416, 416], metavar=('W', 'H'),
                    help="anchors on the letterboxed scale of this input size; --original for original pixels")
    ap.add_argument('--original', action='store_true', help="anchors on the original image scale")
    ap.add_argument('--cluster_num', type=int, default=9)
    ap.add_argument('--seed', type=int, default=None, help="seed of the random start (default: from the OS)")
    ap.add_argument('--output', type=str, default=None, help="write the anchors line (yolo_anchors.txt format) here")
    args = ap.parse_args(sys.argv[1:] if argv is None else argv)
    sizes = parse_anno(args.annotation_path, None if args.original else args.target_size)
    anchors, quality = get_kmeans(sizes, args.cluster_num, rng=np.random.RandomState(args.seed))
    text = ', '.join('{},{}'.format(w, h) for w, h in anchors)
    print('anchors are:')
    print(text)
    print('the average iou is:')
    print(quality)
    if args.output:
        with open(args.output, 'w') as f:
            f.write(text + '\n')
    return anchors, quality


if __name__ == '__main__':
    main()
