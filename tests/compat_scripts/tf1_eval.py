# coding: utf-8
"""Evaluation over an annotation file, written in TF-1 style against the symbols the compat layer provides for the
reference's evaluation script: tf.placeholder (bool / float flags), tf.data.TextLineDataset -> batch -> map(py_func) ->
one-shot iterator, Tensor.set_shape, model.yolov3.forward(is_training=<placeholder>) / compute_loss / predict,
utils.nms_utils.gpu_nms on flag placeholders, tf.train.Saver.restore, utils.eval_utils.{get_preds_gpu, parse_gt_rec,
voc_eval}.  Prints the report and writes it as JSON.

    python -m yolov3_tensorflow_amd.compat.run tests/compat_scripts/tf1_eval.py --eval_file val.txt --restore_path ckpt \
        --anchor_path anchors.txt --json out.json
"""
import argparse
import json

import tensorflow as tf

from model import yolov3
from utils.data_utils import get_batch_data
from utils.eval_utils import get_preds_gpu, parse_gt_rec, voc_eval
from utils.misc_utils import AverageMeter, parse_anchors
from utils.nms_utils import gpu_nms


def input_pipeline(a, anchors):
    """One image per step: annotation line -> (image id, image, three target maps) through the host feeder."""
    kinds = [tf.int64, tf.float32, tf.float32, tf.float32, tf.float32]
    feeder = lambda line: tf.py_func(get_batch_data, [line, a.class_num, a.img_size, anchors, 'val', False, False,
                                                       a.letterbox_resize], kinds)
    ds = tf.data.TextLineDataset(a.eval_file).batch(1).map(feeder, num_parallel_calls=2).prefetch(2)
    ids, image, t13, t26, t52 = ds.make_one_shot_iterator().get_next()
    ids.set_shape([None])
    image.set_shape([None, a.img_size[1], a.img_size[0], 3])
    return ids, image, [t13, t26, t52]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--eval_file', required=True)
    ap.add_argument('--restore_path', required=True)
    ap.add_argument('--anchor_path', required=True)
    ap.add_argument('--json', required=True)
    ap.add_argument('--class_num', type=int, default=80)
    ap.add_argument('--img_size', nargs=2, type=int, default=[416, 416])
    ap.add_argument('--letterbox_resize', type=lambda s: s.lower() == 'true', default=False)
    ap.add_argument('--score_threshold', type=float, default=0.01)
    a = ap.parse_args()
    anchors = parse_anchors(a.anchor_path)
    count = sum(1 for _ in open(a.eval_file))

    training_phase = tf.placeholder(dtype=tf.bool, name='phase_train')
    boxes_in = tf.placeholder(tf.float32, [1, None, None])
    scores_in = tf.placeholder(tf.float32, [1, None, None])
    nms_op = gpu_nms(boxes_in, scores_in, a.class_num, 400, a.score_threshold, 0.45)
    ids, image, targets = input_pipeline(a, anchors)
    net = yolov3(a.class_num, anchors)
    with tf.variable_scope('yolov3'):
        maps = net.forward(image, is_training=training_phase)
    loss = net.compute_loss(maps, targets)
    prediction = net.predict(maps)

    meters = [AverageMeter() for _ in range(5)]
    rows = []
    with tf.Session() as sess:
        sess.run([tf.global_variables_initializer()])
        tf.train.Saver().restore(sess, a.restore_path)
        for _ in range(count):
            got_ids, got_pred, got_loss = sess.run([ids, prediction, loss], feed_dict={training_phase: False})
            rows.extend(get_preds_gpu(sess, nms_op, boxes_in, scores_in, got_ids, got_pred))
            for m, v in zip(meters, got_loss):
                m.update(v)
    truth = parse_gt_rec(a.eval_file, a.img_size, a.letterbox_resize)
    ap_meter = AverageMeter()
    for c in range(a.class_num):
        npos, nd, rec, prec, ap_c = voc_eval(truth, rows, c, iou_thres=0.5, use_07_metric=False)
        ap_meter.update(ap_c, 1)
    report = dict(mAP=float(ap_meter.average), detections=len(rows), loss=[float(m.average) for m in meters])
    print('final mAP: {:.4f}; {} detections; total_loss {:.4f}'.format(report['mAP'], report['detections'], report['loss'][0]))
    with open(a.json, 'w') as f:
        json.dump(report, f)


if __name__ == '__main__':
    main()
