# coding: utf-8
"""Single-image detection on the MI355X path — the counterpart of the reference's test_single_image.py
(same positional/keyword arguments; `--restore_path` takes a darknet .weights file instead of a TF checkpoint,
and the cv2 window/drawing is replaced by printing the detections and an optional PIL-drawn output file).

    python test_single_image.py ./some.jpg --restore_path ./data/darknet_weights/yolov3.weights
"""
from __future__ import division, print_function

import argparse

import numpy as np

import yolov3_tensorflow_amd as y3
from yolov3_tensorflow_amd.utils.data_utils import letterbox_resize
from yolov3_tensorflow_amd.utils.misc_utils import load_weights, parse_anchors, read_class_names, run_ops
from yolov3_tensorflow_amd.utils.nms_utils import gpu_nms

parser = argparse.ArgumentParser(description="YOLO-V3 test single image test procedure.")
parser.add_argument("input_image", type=str, help="The path of the input image.")
parser.add_argument("--anchor_path", type=str, default="./data/yolo_anchors.txt", help="The path of the anchor txt file.")
parser.add_argument("--new_size", nargs='*', type=int, default=[416, 416],
                    help="Resize the input image with `new_size`, size format: [width, height]")
parser.add_argument("--letterbox_resize", type=lambda x: (str(x).lower() == 'true'), default=True,
                    help="Whether to use the letterbox resize.")
parser.add_argument("--class_name_path", type=str, default="./data/coco.names", help="The path of the class names.")
parser.add_argument("--restore_path", type=str, default="./data/darknet_weights/yolov3.weights",
                    help="The path of the darknet weights to restore (random weights if the file is absent).")
parser.add_argument("--output", type=str, default=None, help="Optional path of an annotated output image.")


def main():
    args = parser.parse_args()
    args.anchors = parse_anchors(args.anchor_path)
    args.classes = read_class_names(args.class_name_path)
    args.num_class = len(args.classes)

    from PIL import Image, ImageDraw
    img_ori = np.asarray(Image.open(args.input_image).convert('RGB'))          # RGB (cv2 would give BGR)
    height_ori, width_ori = img_ori.shape[:2]
    if args.letterbox_resize:
        img, resize_ratio, dw, dh = letterbox_resize(img_ori, args.new_size[0], args.new_size[1])
    else:
        sx = np.minimum(np.floor(np.arange(args.new_size[0]) * (width_ori / args.new_size[0])).astype(int), width_ori - 1)
        sy = np.minimum(np.floor(np.arange(args.new_size[1]) * (height_ori / args.new_size[1])).astype(int), height_ori - 1)
        img = img_ori[sy][:, sx]
    img = np.asarray(img, np.float32)
    img = img[np.newaxis, :] / 255.

    yolo_model = y3.yolov3(args.num_class, args.anchors)
    with y3.variable_scope('yolov3'):
        pred_feature_maps = yolo_model.forward(img, False)
        try:
            run_ops(load_weights(y3.global_variables(scope='yolov3'), args.restore_path))
            pred_feature_maps = yolo_model.forward(img, False)
        except (IOError, OSError):
            print('WARNING: %s not found - running with randomly initialised weights' % args.restore_path)
    pred_boxes, pred_confs, pred_probs = yolo_model.predict(pred_feature_maps)
    pred_scores = pred_confs * pred_probs
    boxes_, scores_, labels_ = gpu_nms(pred_boxes, pred_scores, args.num_class, max_boxes=200, score_thresh=0.3,
                                       nms_thresh=0.45)
    boxes_, scores_, labels_ = boxes_.cpu().numpy(), scores_.cpu().numpy(), labels_.cpu().numpy()

    # rescale the coordinates to the original image
    if args.letterbox_resize:
        boxes_[:, [0, 2]] = (boxes_[:, [0, 2]] - dw) / resize_ratio
        boxes_[:, [1, 3]] = (boxes_[:, [1, 3]] - dh) / resize_ratio
    else:
        boxes_[:, [0, 2]] *= (width_ori / float(args.new_size[0]))
        boxes_[:, [1, 3]] *= (height_ori / float(args.new_size[1]))

    print("box coords:")
    print(boxes_)
    print('*' * 30)
    print("scores:")
    print(scores_)
    print('*' * 30)
    print("labels:")
    print(labels_)
    if args.output:
        im = Image.fromarray(img_ori)
        draw = ImageDraw.Draw(im)
        for (x0, y0, x1, y1), s, l in zip(boxes_, scores_, labels_):
            draw.rectangle([float(x0), float(y0), float(x1), float(y1)], outline=(255, 0, 0), width=2)
            draw.text((float(x0) + 2, float(y0) + 2), '%s, %.2f%%' % (args.classes[int(l)], s * 100), fill=(255, 0, 0))
        im.save(args.output)


if __name__ == '__main__':
    main()
