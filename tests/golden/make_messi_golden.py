"""BASELINE configs[0] regression vector: the reference's demo image (data/demo_data/messi.jpg, copied byte for byte to
tests/golden/messi.jpg) through the CPU ORACLE's restatement of test_single_image.py:38-70 —

    decode (PIL, RGB) -> letterbox_resize(416, 416, INTER_NEAREST) -> /255 -> forward -> predict -> conf*prob
    -> per-class tf-NMS (max_boxes 200, nms_thresh 0.45) -> un-letterbox to the 1296x729 frame

with the oracle's synthetic weights (seed 1; no COCO checkpoint can be downloaded here, so the detections are those of
a random network: a REGRESSION vector for the plumbing + numerics, not a statement about Messi).  The score threshold is
the 1 - 150/N quantile of the scores (a random network has no confident detections; stored in the file).

    python tests/golden/make_messi_golden.py        # writes tests/golden/messi_config1_golden.npz (committed)

tests/test_messi_gpu.py runs the PRODUCT path (test_single_image.py twin) on the same file and compares.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326],
                   np.float32).reshape(9, 2)
SIZE = 416


def letterbox_nearest(img, new_w, new_h):
    """utils/data_aug.py:274-293 with cv2.INTER_NEAREST, restated independently of the product's data_utils."""
    h, w = img.shape[:2]
    ratio = min(new_w / w, new_h / h)
    rw, rh = int(ratio * w), int(ratio * h)
    xs = np.minimum((np.arange(rw) * (w / rw)).astype(np.int64), w - 1)      # floor of a non-negative product
    ys = np.minimum((np.arange(rh) * (h / rh)).astype(np.int64), h - 1)
    out = np.full((new_h, new_w, 3), 128, np.uint8)
    dw, dh = int((new_w - rw) / 2), int((new_h - rh) / 2)
    out[dh:dh + rh, dw:dw + rw] = img[ys][:, xs]
    return out, ratio, dw, dh


def main():
    import torch
    from PIL import Image
    from oracle import yolo_ref, nms_ref
    img = np.asarray(Image.open(os.path.join(HERE, 'messi.jpg')).convert('RGB'))
    assert img.shape == (729, 1296, 3), img.shape
    lb, ratio, dw, dh = letterbox_nearest(img, SIZE, SIZE)
    x = (lb.astype(np.float32) / 255.)[None]
    params = yolo_ref.synthetic_params(80, seed=1)
    fms = yolo_ref.forward(params, x, dtype=torch.float32)
    boxes, confs, probs = yolo_ref.predict(fms, ANCHORS, [SIZE, SIZE], 80)
    scores = confs * probs
    thr = float(np.quantile(scores, 1 - 150.0 / scores.size))
    b, s, l, idx = nms_ref.c_per_class('tf', boxes[0], scores[0], 80, 200, thr, 0.45)
    b = b.copy()
    b[:, [0, 2]] = (b[:, [0, 2]] - dw) / ratio
    b[:, [1, 3]] = (b[:, [1, 3]] - dh) / ratio
    out = os.path.join(HERE, 'messi_config1_golden.npz')
    np.savez_compressed(out, score_thresh=np.float64(thr), boxes=b.astype(np.float32), scores=s.astype(np.float32),
                        labels=l.astype(np.int32), index=idx.astype(np.int32), ratio=np.float64(ratio),
                        dw=np.int64(dw), dh=np.int64(dh), letterbox_sum=np.int64(lb.astype(np.int64).sum()),
                        letterbox_crc=np.int64(int(np.bitwise_xor.reduce(lb.astype(np.int64).ravel() * (np.arange(lb.size) % 251 + 1)))),
                        n_candidates=np.int64(int((scores >= thr).sum())))
    print('wrote %s: %d detections from %d candidates at score_thresh %.6g' % (out, len(l), int((scores >= thr).sum()), thr))


if __name__ == '__main__':
    main()
