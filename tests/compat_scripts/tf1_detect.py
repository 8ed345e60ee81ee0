# coding: utf-8
"""A TF-1 style single-image detection script written against the SAME symbols as the reference's
test_single_image.py (cv2.imread / cvtColor, utils.data_aug.letterbox_resize, tf.Session / placeholder /
variable_scope / train.Saver.restore / sess.run(fetches, feed_dict), model.yolov3.forward / predict,
utils.nms_utils.gpu_nms, utils.plot_utils), with the thresholds and an output file on the command line.

    python -m yolov3_tensorflow_amd.compat.run tests/compat_scripts/tf1_detect.py <image> <ckpt> <anchors.txt> \
        <score_thresh> <out.npz> [<annotated.jpg>]
"""
from __future__ import division, print_function

import sys

import numpy as np
import tensorflow as tf
import cv2

from utils.misc_utils import parse_anchors
from utils.nms_utils import gpu_nms
from utils.plot_utils import get_color_table, plot_one_box
from utils.data_aug import letterbox_resize

from model import yolov3

image_path, restore_path, anchor_path, score_thresh, out_path = sys.argv[1:6]
annotated = sys.argv[6] if len(sys.argv) > 6 else None
num_class, new_size = 80, [416, 416]
anchors = parse_anchors(anchor_path)
colors = get_color_table(num_class)

img_ori = cv2.imread(image_path)
img, resize_ratio, dw, dh = letterbox_resize(img_ori, new_size[0], new_size[1])
img = cv2.cvtColor(img, cv2.COLOR_BGR2RGB)
img = np.asarray(img, np.float32)[np.newaxis, :] / 255.

with tf.Session() as sess:
    input_data = tf.placeholder(tf.float32, [1, new_size[1], new_size[0], 3], name='input_data')
    yolo_model = yolov3(num_class, anchors)
    with tf.variable_scope('yolov3'):
        pred_feature_maps = yolo_model.forward(input_data, False)
    pred_boxes, pred_confs, pred_probs = yolo_model.predict(pred_feature_maps)
    pred_scores = pred_confs * pred_probs
    boxes, scores, labels = gpu_nms(pred_boxes, pred_scores, num_class, max_boxes=200,
                                    score_thresh=float(score_thresh), nms_thresh=0.45)
    saver = tf.train.Saver()
    saver.restore(sess, restore_path)
    boxes_, scores_, labels_ = sess.run([boxes, scores, labels], feed_dict={input_data: img})

    boxes_[:, [0, 2]] = (boxes_[:, [0, 2]] - dw) / resize_ratio
    boxes_[:, [1, 3]] = (boxes_[:, [1, 3]] - dh) / resize_ratio
    print('%d detections' % len(labels_))
    np.savez(out_path, boxes=boxes_, scores=scores_, labels=labels_)
    if annotated:
        for i in range(min(len(boxes_), 20)):
            plot_one_box(img_ori, boxes_[i], label='%d, %.2f%%' % (labels_[i], scores_[i] * 100), color=colors[labels_[i]])
        cv2.imshow('Detection result', img_ori)
        cv2.imwrite(annotated, img_ori)
        cv2.waitKey(0)
