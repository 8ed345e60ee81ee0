# coding: utf-8
"""Darknet weights -> checkpoint, written in TF-1 style against the symbols the compat layer provides
(tf.Session / placeholder / variable_scope / global_variables / train.Saver, model.yolov3,
utils.misc_utils.load_weights) - the same ones the reference's conversion script touches, exercised here with explicit
paths and a few self-checks.

    python -m yolov3_tensorflow_amd.compat.run tests/compat_scripts/tf1_convert.py <in.weights> <out.ckpt> <anchors.txt>
"""
import argparse

import tensorflow as tf

from model import yolov3
from utils.misc_utils import load_weights, parse_anchors

SCOPE = 'yolov3'


def build(class_count, anchor_file, side):
    """Graph of one forward pass; returns the network's variables in creation order."""
    network = yolov3(class_count, parse_anchors(anchor_file))
    image = tf.placeholder(tf.float32, [1, side, side, 3], name='image')
    with tf.variable_scope(SCOPE):
        maps = network.forward(image)
    assert len(maps) == 3
    return tf.global_variables(scope=SCOPE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('weights')
    ap.add_argument('checkpoint')
    ap.add_argument('anchors')
    ap.add_argument('--classes', type=int, default=80)
    ap.add_argument('--side', type=int, default=416)
    a = ap.parse_args()
    session = tf.Session()
    variables = build(a.classes, a.anchors, a.side)
    assign_ops = load_weights(variables, a.weights)
    session.run(tf.global_variables_initializer())
    session.run(assign_ops)
    written = tf.train.Saver(var_list=variables).save(session, save_path=a.checkpoint)
    session.close()
    print('checkpoint saved to {} ({} variables)'.format(written, len(variables)))


if __name__ == '__main__':
    main()
