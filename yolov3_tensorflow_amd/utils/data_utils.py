# coding: utf-8
"""The slice of the reference's utils/data_utils.py / utils/data_aug.py that sits immediately either side of
the hot path: annotation-line parsing (`parse_line`), target assignment (`process_box`,
on the device, batched) and the OpenCV-free letterbox used by the single-image and eval paths.  Augmentation
stays out of scope."""
from __future__ import division, print_function

import ctypes

import numpy as np
import torch

from .. import _lib
from .. import framework as fw


def parse_line(line):
    '''
    Given a line from the training/test txt file, return parsed info (reference utils/data_utils.py:15-48).
    line format: line_index, img_path, img_width, img_height, [box_info_1 (5 number)], ...
    return:
        line_idx: int
        pic_path: string.
        boxes: shape [N, 4] float32, [x_min, y_min, x_max, y_max] per ground-truth object
        labels: shape [N] int64, class index.
        img_width: int.
        img_height: int
    '''
    if isinstance(line, bytes):
        line = line.decode()
    fields = line.strip().split(' ')
    assert len(fields) > 8, ('Annotation error! Please check your annotation file. Make sure there is at least one '
                             'target object in each image.')
    line_idx, pic_path = int(fields[0]), fields[1]
    img_width, img_height = int(fields[2]), int(fields[3])
    obj = fields[4:]
    assert len(obj) % 5 == 0, ('Annotation error! Please check your annotation file. Maybe partially missing some '
                               'coordinates?')
    table = np.array(obj, dtype=object).reshape(-1, 5)
    labels = np.asarray([int(v) for v in table[:, 0]], np.int64)
    boxes = np.asarray([[float(v) for v in row] for row in table[:, 1:]], np.float32).reshape(-1, 4)
    return line_idx, pic_path, boxes, labels, img_width, img_height


def process_box(boxes, labels, img_size, class_num, anchors):
    '''
    Generate the y_true label, i.e. the ground truth feature_maps in 3 different scales
    (reference utils/data_utils.py:51-115), for ONE image, on the device.
    params:
        boxes: [N, 5] shape, float32 dtype. `x_min, y_min, x_max, y_mix, mixup_weight`.
        labels: [N] shape, int64 dtype.
        img_size: [width, height]
        class_num: int.
        anchors: [9, 2] shape, float32 dtype.
    returns y_true_13, y_true_26, y_true_52 as numpy arrays (like the reference).
    '''
    boxes = np.asarray(boxes, np.float32).reshape(1, -1, 5)
    labels = np.asarray(labels).reshape(1, -1)
    ys = process_box_batch(boxes, labels, [boxes.shape[1]], img_size, class_num, anchors)
    return tuple(y[0].cpu().numpy() for y in ys)


def process_box_batch(boxes, labels, counts, img_size, class_num, anchors):
    """Batched device form: boxes [N,Kmax,5], labels [N,Kmax], counts [N] -> three device tensors
    [N,g,g,3,6+C] ready for yolov3.compute_loss (no host round trip per image)."""
    b = fw.as_device_f32(boxes)
    if b.dim() != 3 or b.shape[2] != 5:
        raise ValueError("boxes must be [N, Kmax, 5]")
    n, kmax, _ = b.shape
    dev = b.device
    lab = torch.as_tensor(np.asarray(labels), dtype=torch.int32).reshape(n, kmax).to(dev).contiguous()
    cnt = torch.as_tensor(np.asarray(counts), dtype=torch.int32).reshape(n).to(dev).contiguous()
    w, h = int(img_size[0]), int(img_size[1])
    C = int(class_num)
    anc = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(9, 2))
    ys = [torch.empty((n, h // s, w // s, 3, 6 + C), dtype=torch.float32, device=dev) for s in (32, 16, 8)]
    _lib.check(_lib.lib().y3_process_box(fw.context(dev), fw.ptr(b), fw.ptr(lab), fw.ptr(cnt), n, kmax, C, w, h,
                                         anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                         fw.ptr(ys[0]), fw.ptr(ys[1]), fw.ptr(ys[2])))
    return ys


def letterbox_resize(img, new_width, new_height, interp=0):
    '''
    Letterbox resize keeping the aspect ratio (reference utils/data_aug.py:274-293), without OpenCV.
    img: HxWx3 uint8.  interp 0 = cv2.INTER_NEAREST semantics: src = min(floor(dst * src_size / dst_size),
    src_size - 1) (no half-pixel centre); other interpolations are not provided.
    returns image_padded, resize_ratio, dw, dh
    '''
    if interp != 0:
        raise ValueError("only interp=0 (nearest, the value the demo/eval call sites use) is implemented")
    img = np.asarray(img)
    ori_height, ori_width = img.shape[:2]
    resize_ratio = min(new_width / ori_width, new_height / ori_height)
    resize_w = int(resize_ratio * ori_width)
    resize_h = int(resize_ratio * ori_height)
    sx = np.minimum(np.floor(np.arange(resize_w) * (ori_width / resize_w)).astype(np.int64), ori_width - 1)
    sy = np.minimum(np.floor(np.arange(resize_h) * (ori_height / resize_h)).astype(np.int64), ori_height - 1)
    resized = img[sy][:, sx]
    image_padded = np.full((new_height, new_width, 3), 128, np.uint8)
    dw = int((new_width - resize_w) / 2)
    dh = int((new_height - resize_h) / 2)
    image_padded[dh: resize_h + dh, dw: resize_w + dw, :] = resized
    return image_padded, resize_ratio, dw, dh
