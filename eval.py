# coding: utf-8
"""Twin of the reference's eval.py (same command line, same printed report) on the MI355X-native path.

What changes underneath: the reference evaluates ONE image per `sess.run`, pulls the decoded
predictions to the host and feeds them back for a second `sess.run` of the NMS op; here a batch of `--batch_size`
images goes forward -> decode -> per-class NMS on the device in one launch set (`yolov3.detect` building blocks),
the loss of the batch is computed on the device from `process_box_batch` targets, and only the surviving detections
cross to the host.  Weights: a darknet `.weights` file (`--restore_path`; the reference restores a TF checkpoint
converted from the same file) or a native `.npz` checkpoint written by train.py / convert_weight.py.  Images are read
with PIL and resized like the reference's validation path (cv2.INTER_LINEAR arithmetic, plain or letterboxed) by the
feeder's worker threads (yolov3_tensorflow_amd.feeder, 'val' mode), ahead of the device.
"""
from __future__ import division, print_function

import argparse
import sys


def _flag(text):
    return str(text).lower() == 'true'


# (option, type, default, help).  The first twelve are the reference's options (eval.py:22-60) with its defaults, except
# restore_path, which names a darknet .weights file or a native .npz instead of a TF checkpoint.
OPTIONS = (
    ('eval_file', str, './data/my_data/val.txt', 'annotation txt file of the validation / test set'),
    ('restore_path', str, './data/darknet_weights/yolov3.weights', 'darknet .weights file or native .npz checkpoint'),
    ('anchor_path', str, './data/yolo_anchors.txt', 'anchor txt file'),
    ('class_name_path', str, './data/coco.names', 'class names file'),
    ('img_size', int, [416, 416], 'network input size as: width height'),
    ('letterbox_resize', _flag, False, 'true: keep the aspect ratio and pad; false: plain stretch'),
    ('num_threads', int, 10, 'decode / resize worker threads of the feeder'),
    ('prefetech_buffer', int, 5, 'batches the feeder keeps ahead of the device'),
    ('nms_threshold', float, 0.45, 'IoU above which NMS suppresses a box'),
    ('score_threshold', float, 0.01, 'class score below which a box is not a candidate'),
    ('nms_topk', int, 400, 'most detections kept per class'),
    ('use_voc_07_metric', _flag, False, 'true: the 11-point VOC 2007 AP'),
    ('batch_size', int, 32, 'images per device batch (the reference evaluates one at a time)'),
    ('compute_dtype', str, 'f32_wino', 'f32_wino (exact fp32, Winograd 3x3 kernels) | f32 | f32_bf16x6 (docs/history_r01_r05.md 4.3-4.4)'),
)


def build_parser():
    parser = argparse.ArgumentParser(description="YOLO-V3 eval procedure.")
    for name, kind, default, text in OPTIONS:
        extra = {'nargs': '*'} if isinstance(default, list) else {}
        parser.add_argument('--' + name, type=kind, default=default, help=text, **extra)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.feeder import Feeder
    from yolov3_tensorflow_amd.utils.eval_utils import get_preds_batch, voc_eval, parse_gt_rec
    from yolov3_tensorflow_amd.utils import eval_utils
    from yolov3_tensorflow_amd.utils.misc_utils import (parse_anchors, read_class_names, AverageMeter, load_weights,
                                                        run_ops, Saver)
    from yolov3_tensorflow_amd.utils.nms_utils import gpu_nms_batched

    args.anchors = parse_anchors(args.anchor_path)
    args.classes = read_class_names(args.class_name_path)
    args.class_num = len(args.classes)
    lines = [l for l in open(args.eval_file, 'r').readlines() if l.strip()]
    args.img_cnt = len(lines)

    yolo_model = y3.yolov3(args.class_num, args.anchors)
    yolo_model.compute_dtype = args.compute_dtype
    with y3.variable_scope('yolov3'):
        yolo_model.forward(torch.zeros((1, 64, 64, 3)), False)            # create the variables
        if args.restore_path.endswith('.npz'):       # native checkpoint (train.py best_model_*.npz, convert_weight.py)
            Saver(y3.global_variables(scope='yolov3')).restore(args.restore_path)
        else:
            run_ops(load_weights(y3.global_variables(scope='yolov3'), args.restore_path))

    print('\n----------- start to eval -----------\n')
    meters = [AverageMeter() for _ in range(5)]      # total, xy, wh, conf, class
    val_preds = []
    # the reference's tf.data pipeline (eval.py:54-62: batch, py_func(get_batch_data) on num_threads workers, prefetch) is
    # the feeder in 'val' mode: decode + resize (cv2.INTER_LINEAR arithmetic, plain or letterboxed) on worker threads into
    # pinned buffers, side-stream upload and target assignment on the device, overlapped with the forward of the batch before
    feeder = Feeder(lines, args.batch_size, args.class_num, args.img_size, args.anchors, mode='val',
                    letterbox_resize=args.letterbox_resize, num_threads=args.num_threads, prefetch=args.prefetech_buffer)
    def consume(done):
        # host-side bookkeeping of a batch whose device work was enqueued one batch ago: by now its counts have usually
        # landed, so the device never waits for the host between two batches (the reference ran two sess.run per IMAGE)
        ids, dets, loss = done
        val_preds.extend(get_preds_batch(ids, dets))
        for m, v in zip(meters, loss):
            m.update(float(v), len(ids))

    in_flight = None
    for batch in feeder.epoch(0):
        with y3.variable_scope('yolov3'):
            fms = yolo_model.forward(batch.images, False)
        loss = yolo_model.compute_loss(fms, batch.y_true)
        pb, _, _, ps = yolo_model.predict(fms, with_scores=True)
        dets = gpu_nms_batched(pb, ps, args.class_num, args.nms_topk, args.score_threshold, args.nms_threshold, lazy=True)
        if in_flight is not None:
            consume(in_flight)
        in_flight = (batch.image_ids, dets, loss)
    if in_flight is not None:
        consume(in_flight)
    feeder.close()

    rec_total, prec_total, ap_total = AverageMeter(), AverageMeter(), AverageMeter()
    eval_utils.gt_dict = {}
    gt_dict = parse_gt_rec(args.eval_file, args.img_size, args.letterbox_resize)
    print('mAP eval:')
    for ii in range(args.class_num):
        npos, nd, rec, prec, ap = voc_eval(gt_dict, val_preds, ii, iou_thres=0.5,
                                           use_07_metric=args.use_voc_07_metric)
        rec_total.update(rec, npos)
        prec_total.update(prec, nd)
        ap_total.update(ap, 1)
        print('Class {}: Recall: {:.4f}, Precision: {:.4f}, AP: {:.4f}'.format(ii, rec, prec, ap))

    mAP = ap_total.average
    print('final mAP: {:.4f}'.format(mAP))
    print("recall: {:.3f}, precision: {:.3f}".format(rec_total.average, prec_total.average))
    print("total_loss: {:.3f}, loss_xy: {:.3f}, loss_wh: {:.3f}, loss_conf: {:.3f}, loss_class: {:.3f}".format(
        *[m.average for m in meters]))
    return {'mAP': mAP, 'val_preds': val_preds, 'loss': [m.average for m in meters], 'gt_dict': gt_dict}


if __name__ == '__main__':
    main(sys.argv[1:])
