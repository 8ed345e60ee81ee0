# coding: utf-8
"""A TF-1 style weight conversion script written against the SAME symbols as the reference's convert_weight.py
(tf.Session / placeholder / variable_scope / global_variables / train.Saver, `from model import yolov3`,
utils.misc_utils.load_weights), with the paths on the command line.  Run through the compat layer:

    python -m yolov3_tensorflow_amd.compat.run tests/compat_scripts/tf1_convert.py <in.weights> <out.ckpt> <anchors.txt>
"""
from __future__ import division, print_function

import sys

import tensorflow as tf

from model import yolov3
from utils.misc_utils import parse_anchors, load_weights

weight_path, save_path, anchor_path = sys.argv[1:4]
anchors = parse_anchors(anchor_path)

model = yolov3(80, anchors)
with tf.Session() as sess:
    inputs = tf.placeholder(tf.float32, [1, 416, 416, 3])
    with tf.variable_scope('yolov3'):
        feature_map = model.forward(inputs)
    saver = tf.train.Saver(var_list=tf.global_variables(scope='yolov3'))
    load_ops = load_weights(tf.global_variables(scope='yolov3'), weight_path)
    sess.run(load_ops)
    saver.save(sess, save_path=save_path)
    print('checkpoint saved to {} ({} variables)'.format(save_path, len(tf.global_variables(scope='yolov3'))))
