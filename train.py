# coding: utf-8
"""Twin of the reference's train.py on the MI355X-native path (BASELINE configs[3]).

Same training recipe and the same names as the reference's args.py (every variable of args.py is a command-line
option with the reference's default): restore (darknet weights, with `restore_include` / `restore_exclude` scope
filters), `update_part` fine-tuning, warm-up + the five learning-rate schedules, the four optimizers, per-tensor
clip_by_norm(100), L2 weight decay, label smoothing / focal loss, periodic recall/precision on the training batch
(`evaluate_on_gpu`), periodic mAP on the validation file (`voc_eval`), multi-scale training.  One `Trainer.step` =
forward(is_training=True) -> compute_loss -> backward -> clip -> update, all on the device; under
`python -m torch.distributed.run --nproc-per-node N train.py ...` every rank trains on its shard of each batch and
the flat gradient buffer is all-reduced over RCCL.

The batches come from yolov3_tensorflow_amd.feeder.Feeder = the reference's tf.data pipeline (train.py:34-52:
shuffle -> batch -> get_batch_data on `num_threads` threads -> prefetch): decode, the augmentation chain of
utils/data_aug.py (colour distortion, expansion, constrained crop, random-interpolation resize, flip, mix-up), resize and
the host-to-device copy of the next `prefetech_buffer` batches run under the current step; targets are assigned on the
device.  What it does not have: TensorBoard summaries and TF checkpoints - weights are saved as darknet `.weights` files
and native `.npz` checkpoints.
"""
from __future__ import division, print_function

import argparse
import logging
import math
import os
import random
import sys

import numpy as np


def _bool(x):
    return str(x).lower() == 'true'


def _scopes(x):
    return None if x in (None, '', 'None', 'none') else [s for s in x.split(',') if s]


def build_parser():
    p = argparse.ArgumentParser(description="YOLO-V3 training procedure (options = the variables of args.py).")
    # paths
    p.add_argument('--train_file', default='./data/my_data/train.txt')
    p.add_argument('--val_file', default='./data/my_data/val.txt')
    p.add_argument('--restore_path', default='./data/darknet_weights/yolov3.weights',
                   help="darknet .weights file or native .npz checkpoint to start from ('' = random initialisation)")
    p.add_argument('--save_optimizer', type=_bool, default=True,
                   help="also save native .npz checkpoints with the optimizer slots and global_step")
    p.add_argument('--save_dir', default='./checkpoint/')
    p.add_argument('--progress_log_path', default='./data/progress.log')
    p.add_argument('--anchor_path', default='./data/yolo_anchors.txt')
    p.add_argument('--class_name_path', default='./data/coco.names')
    # training numbers
    p.add_argument('--batch_size', type=int, default=6)
    p.add_argument('--img_size', nargs=2, type=int, default=[416, 416], help='[width, height]')
    p.add_argument('--letterbox_resize', type=_bool, default=True)
    p.add_argument('--total_epoches', type=int, default=100)
    p.add_argument('--train_evaluation_step', type=int, default=100)
    p.add_argument('--val_evaluation_epoch', type=int, default=2)
    p.add_argument('--save_epoch', type=int, default=10)
    p.add_argument('--batch_norm_decay', type=float, default=0.99)
    p.add_argument('--weight_decay', type=float, default=5e-4)
    p.add_argument('--global_step', type=int, default=0)
    # learning rate and optimizer
    p.add_argument('--optimizer_name', default='momentum', help='sgd | momentum | adam | rmsprop')
    p.add_argument('--learning_rate_init', type=float, default=1e-4)
    p.add_argument('--lr_type', default='piecewise',
                   help='fixed | exponential | cosine_decay | cosine_decay_restart | piecewise')
    p.add_argument('--lr_decay_epoch', type=float, default=5)
    p.add_argument('--lr_decay_factor', type=float, default=0.96)
    p.add_argument('--lr_lower_bound', type=float, default=1e-6)
    p.add_argument('--pw_boundaries', nargs='*', type=float, default=[30, 50], help='epoch based boundaries')
    p.add_argument('--pw_values', nargs='*', type=float, default=None,
                   help='default [learning_rate_init, 3e-5, 1e-5]')
    # load and finetune
    p.add_argument('--restore_include', type=_scopes, default=None, help='comma separated scopes, or None')
    p.add_argument('--restore_exclude', type=_scopes,
                   default=['yolov3/yolov3_head/Conv_14', 'yolov3/yolov3_head/Conv_6', 'yolov3/yolov3_head/Conv_22'])
    p.add_argument('--update_part', type=_scopes, default=['yolov3/yolov3_head'],
                   help="comma separated scopes to train; 'None' = the whole model")
    # tf.data parameters (args.py:33-34)
    p.add_argument('--num_threads', type=int, default=10, help='threads decoding / augmenting / resizing images')
    p.add_argument('--prefetech_buffer', type=int, default=5, help='batches prepared ahead of the train step')
    p.add_argument('--feeder_backend', choices=['thread', 'process'], default=None,
                   help='feeder workers: threads (default; about 1,000 images/s per GPU) or processes filling page-locked '
                        'shared memory (past one interpreter, e.g. 32 workers: 3,300 images/s)')
    # other strategies
    p.add_argument('--use_mix_up', type=_bool, default=True)
    p.add_argument('--augment', type=_bool, default=True,
                   help="false: feed the training images through the validation preprocessing (plain resize, no "
                        "augmentation, no mix-up) - not a reference option; used by the memorisation test")
    p.add_argument('--fix_crop_labels', type=_bool, default=False,
                   help="true: a box keeps its own class when the random crop drops an earlier box of the image (the "
                        "reference filters the boxes but not their labels, utils/data_utils.py:150-153; false reproduces it)")
    p.add_argument('--multi_scale_train', type=_bool, default=True)
    p.add_argument('--use_label_smooth', type=_bool, default=True)
    p.add_argument('--use_focal_loss', type=_bool, default=True)
    p.add_argument('--use_warm_up', type=_bool, default=True)
    p.add_argument('--warm_up_epoch', type=float, default=3)
    # validation constants
    p.add_argument('--nms_threshold', type=float, default=0.45)
    p.add_argument('--score_threshold', type=float, default=0.01)
    p.add_argument('--nms_topk', type=int, default=150)
    p.add_argument('--eval_threshold', type=float, default=0.5)
    p.add_argument('--use_voc_07_metric', type=_bool, default=False)
    p.add_argument('--seed', type=int, default=0)
    return p


def variables_to_restore(variables, include, exclude):
    """tf.contrib.framework.get_variables_to_restore(include, exclude) on our variable list (train.py:81)."""
    from yolov3_tensorflow_amd.utils.misc_utils import get_variables_to_restore
    return get_variables_to_restore(variables, include, exclude)


def validate(model, y3, args, lines):
    """mAP / recall / precision / losses over the validation file (train.py:171-214), batched on the device."""
    from yolov3_tensorflow_amd.utils import eval_utils
    from yolov3_tensorflow_amd.utils.misc_utils import AverageMeter
    from yolov3_tensorflow_amd.utils.nms_utils import gpu_nms_batched
    from yolov3_tensorflow_amd.feeder import Feeder
    meters = [AverageMeter() for _ in range(5)]
    val_preds = []
    feeder = Feeder(lines, args.batch_size, args.class_num, args.img_size, args.anchors, mode='val',
                    letterbox_resize=args.letterbox_resize, num_threads=args.num_threads, prefetch=args.prefetech_buffer,
                    backend=getattr(args, 'feeder_backend', None))
    for batch in feeder.epoch(0):
        with y3.variable_scope('yolov3'):
            fms = model.forward(batch.images, False)
        loss = model.compute_loss(fms, batch.y_true)
        pb, _, _, ps = model.predict(fms, with_scores=True)
        dets = gpu_nms_batched(pb, ps, args.class_num, args.nms_topk, args.score_threshold, args.nms_threshold)
        val_preds.extend(eval_utils.get_preds_batch(batch.image_ids, dets))
        for m, v in zip(meters, loss):
            m.update(float(v), len(batch.image_ids))
    eval_utils.gt_dict = {}
    gt = eval_utils.parse_gt_rec(args.val_file, args.img_size, args.letterbox_resize)
    rec_total, prec_total, ap_total = AverageMeter(), AverageMeter(), AverageMeter()
    info = ''
    for ii in range(args.class_num):
        npos, nd, rec, prec, ap = eval_utils.voc_eval(gt, val_preds, ii, iou_thres=args.eval_threshold,
                                                      use_07_metric=args.use_voc_07_metric)
        info += 'EVAL: Class {}: Recall: {:.4f}, Precision: {:.4f}, AP: {:.4f}\n'.format(ii, rec, prec, ap)
        rec_total.update(rec, npos)
        prec_total.update(prec, nd)
        ap_total.update(ap, 1)
    info += 'EVAL: Recall: {:.4f}, Precison: {:.4f}, mAP: {:.4f}\n'.format(rec_total.average, prec_total.average,
                                                                          ap_total.average)
    info += 'EVAL: loss: total: {:.2f}, xy: {:.2f}, wh: {:.2f}, conf: {:.2f}, class: {:.2f}\n'.format(
        *[m.average for m in meters])
    return ap_total.average, rec_total.average, prec_total.average, [m.average for m in meters], info


def main(argv=None):
    args = build_parser().parse_args(argv)
    os.environ['Y3_FIX_CROP_LABELS'] = '1' if args.fix_crop_labels else '0'      # (read by the feeder's workers too)
    import functools
    import torch
    import torch.distributed as dist
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import training, framework as fw
    from yolov3_tensorflow_amd.utils.eval_utils import evaluate_on_gpu
    from yolov3_tensorflow_amd.utils.misc_utils import (parse_anchors, read_class_names, AverageMeter,
                                                        config_learning_rate, config_optimizer, load_weights,
                                                        save_weights, run_ops, Saver)
    from yolov3_tensorflow_amd.utils.nms_utils import gpu_nms

    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        y3.set_default_device('cuda:%d' % local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

    # ---- the derived parameters of args.py:79-86 ----
    args.anchors = parse_anchors(args.anchor_path)
    args.classes = read_class_names(args.class_name_path)
    args.class_num = len(args.classes)
    train_lines = [l for l in open(args.train_file, 'r').readlines() if l.strip()]
    val_lines = [l for l in open(args.val_file, 'r').readlines() if l.strip()]
    args.train_img_cnt, args.val_img_cnt = len(train_lines), len(val_lines)
    args.train_batch_num = int(math.ceil(float(args.train_img_cnt) / args.batch_size))
    args.lr_decay_freq = int(args.train_batch_num * args.lr_decay_epoch)
    if args.pw_values is None:
        args.pw_values = [args.learning_rate_init, 3e-5, 1e-5]
    args.pw_boundaries = [float(i) * args.train_batch_num + args.global_step for i in args.pw_boundaries]

    log = logging.getLogger('yolo355.train')
    log.setLevel(logging.DEBUG)
    if rank == 0 and args.progress_log_path:
        os.makedirs(os.path.dirname(os.path.abspath(args.progress_log_path)), exist_ok=True)
        handler = logging.FileHandler(args.progress_log_path, mode='w')
        handler.setFormatter(logging.Formatter('%(asctime)s %(levelname)s %(message)s', '%a, %d %b %Y %H:%M:%S'))
        log.handlers = [handler]

    random.seed(args.seed)
    y3.set_init_seed(args.seed)
    yolo_model = y3.yolov3(args.class_num, args.anchors, args.use_label_smooth, args.use_focal_loss,
                           args.batch_norm_decay, args.weight_decay, use_static_shape=False)
    with y3.variable_scope('yolov3'):
        yolo_model.forward(torch.zeros((1, 64, 64, 3)), False)                 # create the variables
    variables = y3.global_variables(scope='yolov3')
    resumed_step = None
    to_restore = variables_to_restore(variables, args.restore_include, args.restore_exclude)
    if args.restore_path.endswith('.npz'):
        resumed_step = Saver(to_restore).restore(args.restore_path)
    elif args.restore_path:
        restore = set(v.op_name for v in to_restore)
        run_ops([op for op in load_weights(variables, args.restore_path) if op.var.op_name in restore])
    update_vars = None if args.update_part is None else variables_to_restore(variables, args.update_part, None)

    # learning rate: warm-up ramp, then the configured schedule shifted by the warm-up length (train.py:93-99)
    warm = args.train_batch_num * args.warm_up_epoch

    def learning_rate(global_step):
        if args.use_warm_up and global_step < warm:
            return args.learning_rate_init * global_step / warm
        return config_learning_rate(args, global_step - warm if args.use_warm_up else global_step)

    optimizer = config_optimizer(args.optimizer_name, learning_rate)
    if args.restore_path.endswith('.npz') and args.save_optimizer:
        # optimizer slots + step only, and only for variables that were restored: restore_exclude stays excluded
        restored = set(v.op_name for v in to_restore)
        Saver([v for v in (update_vars if update_vars is not None else variables) if v.op_name in restored]).restore(
            args.restore_path, optimizer, variables=False)
        if resumed_step is not None and args.global_step == 0:
            args.global_step = int(resumed_step)
    trainer = training.Trainer(yolo_model, optimizer, update_vars=update_vars, global_step=float(args.global_step),
                               process_group=dist.group.WORLD if world > 1 else None)
    gpu_nms_op = functools.partial(gpu_nms, num_classes=args.class_num, max_boxes=args.nms_topk,
                                   score_thresh=args.score_threshold, nms_thresh=args.nms_threshold)
    if rank == 0:
        print('\n----------- start to train -----------\n')
    best_mAP = -np.inf
    history = {'loss': [], 'recall': [], 'mAP': [], 'global_step_start': int(trainer.global_step)}
    from yolov3_tensorflow_amd.feeder import Feeder
    if not args.augment and (args.multi_scale_train or args.use_mix_up) and rank == 0:
        # ('val' mode has neither multi-scale sizes nor mix-up - say so instead of dropping them silently)
        print("train.py: --augment false feeds the network in 'val' mode: --multi_scale_train and --use_mix_up are OFF for this run",
              file=sys.stderr)
    feeder = Feeder(train_lines, args.batch_size, args.class_num, args.img_size, args.anchors,
                    mode='train' if args.augment else 'val', shuffle=True,
                    multi_scale=args.multi_scale_train, use_mix_up=args.use_mix_up, letterbox_resize=args.letterbox_resize,
                    num_threads=args.num_threads, prefetch=args.prefetech_buffer, seed=args.seed, rank=rank, world=world,
                    backend=args.feeder_backend)
    for epoch in range(args.total_epoches):
        meters = [AverageMeter() for _ in range(5)]
        for i, batch in enumerate(feeder.epoch(epoch)):
            images, y_true = batch.images, batch.y_true
            with y3.variable_scope('yolov3'):
                loss = trainer.step(images, y_true)
            n = len(batch.image_ids)
            for m, v in zip(meters, loss):
                m.update(float(v), n)
            fw.check_context()      # (the loss read-out synchronised) a device-side failure of this step raises here
            history['loss'].append(float(loss[0]))
            gs, lr = trainer.global_step, learning_rate(trainer.global_step - 1)
            if gs % args.train_evaluation_step == 0 and gs > 0:
                with y3.variable_scope('yolov3'):
                    y_pred = yolo_model.predict(yolo_model.forward(images, False))
                recall, precision = evaluate_on_gpu(None, gpu_nms_op, None, None, y_pred, y_true, args.class_num,
                                                    args.nms_threshold)
                history['recall'].append(recall)
                info = "Epoch: {}, global_step: {} | loss: total: {:.2f}, xy: {:.2f}, wh: {:.2f}, conf: {:.2f}, " \
                       "class: {:.2f} | ".format(epoch, int(gs), *[m.average for m in meters])
                info += 'Last batch: rec: {:.3f}, prec: {:.3f} | lr: {:.5g}'.format(recall, precision, lr)
                if rank == 0:
                    print(info)
                    log.info(info)
                if np.isnan(meters[0].average):
                    print('****' * 10)
                    raise ArithmeticError(
                        'Gradient exploded! Please train again and you may need modify some parameters.')
        if rank == 0 and epoch % args.save_epoch == 0 and epoch > 0 and meters[0].average <= 2.:
            os.makedirs(args.save_dir, exist_ok=True)
            save_weights(variables, os.path.join(args.save_dir, 'model-epoch_{}_step_{}_loss_{:.4f}_lr_{:.5g}.weights'
                                                 .format(epoch, int(trainer.global_step), meters[0].average, lr)))
        if epoch % args.val_evaluation_epoch == 0 and epoch >= args.warm_up_epoch:
            mAP, rec, prec, vloss, info = validate(yolo_model, y3, args, val_lines)
            history['mAP'].append(mAP)
            info = '======> Epoch: {}, global_step: {}, lr: {:.6g} <======\n'.format(epoch, trainer.global_step, lr) + info
            if rank == 0:
                print(info)
                log.info(info)
                if mAP > best_mAP:
                    best_mAP = mAP
                    os.makedirs(args.save_dir, exist_ok=True)
                    stem = os.path.join(args.save_dir, 'best_model_Epoch_{}_step_{}_mAP_{:.4f}_loss_{:.4f}_lr_{:.7g}'
                                        .format(epoch, int(trainer.global_step), best_mAP, vloss[0], lr))
                    save_weights(variables, stem + '.weights')
                    if args.save_optimizer:
                        Saver(variables).save(stem, optimizer=optimizer, global_step=trainer.global_step)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return history


if __name__ == '__main__':
    main(sys.argv[1:])
