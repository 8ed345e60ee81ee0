# coding: utf-8
"""utils.data_utils of the reference (ref: utils/data_utils.py:17-235): this package's parse_line / process_box /
resize functions under that name, plus the NON-augmenting feeder the evaluation script maps over its tf.data pipeline
(`get_batch_data(..., mode='val')`: read -> resize_with_bbox(INTER_LINEAR) -> RGB / 255 -> process_box).  The
training-time augmentations are out of scope."""
import numpy as np

import cv2

from yolov3_tensorflow_amd.utils.data_utils import *          # noqa: F401,F403
from yolov3_tensorflow_amd.utils.data_utils import parse_line, process_box, resize_with_bbox
from yolov3_tensorflow_amd import compat as _compat


def parse_data(line, class_num, img_size, anchors, mode, letterbox_resize):
    """One annotation line -> (image index, [h,w,3] float32 RGB in [0,1], y_true_13, y_true_26, y_true_52)."""
    if str(mode) == 'train':
        raise NotImplementedError("compat feeder: mode='train' (augmentation, mix-up) is not provided; use train.py "
                                  "of this package")
    if not isinstance(line, (str, bytes)):
        line = line[0] if np.ndim(line) else line.item()
    img_idx, pic_path, boxes, labels, _, _ = parse_line(line)
    img = cv2.imread(pic_path)
    if img is None:
        raise IOError("cannot read image %r" % (pic_path,))
    img, boxes = resize_with_bbox(img, boxes, img_size[0], img_size[1], interp=1, letterbox=letterbox_resize)
    img = cv2.cvtColor(img, cv2.COLOR_BGR2RGB).astype(np.float32) / 255.
    if _compat.dry_run():
        # the target assignment runs on the device (y3_process_box); a dry run only needs tensors of the right shape
        w, h = img_size
        y13, y26, y52 = (np.zeros((h // s, w // s, 3, 6 + class_num), np.float32) for s in (32, 16, 8))
    else:
        weighted = np.concatenate([boxes, np.ones((len(boxes), 1), np.float32)], axis=1)       # mix-up weight 1
        y13, y26, y52 = process_box(weighted, labels, img_size, class_num, anchors)
    return img_idx, img, y13, y26, y52


def get_batch_data(batch_line, class_num, img_size, anchors, mode, multi_scale=False, mix_up=False,
                   letterbox_resize=True, interval=10):
    """A batch of annotation lines -> (image ids int64 [B], images [B,h,w,3], y_true_13, y_true_26, y_true_52)."""
    if isinstance(mode, bytes):
        mode = mode.decode()
    img_size = [int(v) for v in img_size]
    anchors = np.asarray(anchors, np.float32)
    cols = ([], [], [], [], [])
    for line in batch_line:
        for col, val in zip(cols, parse_data(line, int(class_num), img_size, anchors, mode, bool(letterbox_resize))):
            col.append(val)
    ids, imgs, y13, y26, y52 = cols
    return (np.asarray(ids, np.int64), np.asarray(imgs, np.float32), np.asarray(y13, np.float32),
            np.asarray(y26, np.float32), np.asarray(y52, np.float32))
