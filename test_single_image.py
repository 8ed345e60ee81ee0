# coding: utf-8
"""Detect the objects of one image file on the MI355X-native path.

Command line of the reference's test_single_image.py (positional image path, --anchor_path, --new_size,
--letterbox_resize, --class_name_path, --restore_path), with two differences forced by the environment: the weights
come from a darknet `.weights` file (or a native `.npz` checkpoint) instead of a TF checkpoint, and instead of an
OpenCV window the detections are printed and, with --output, drawn into an image file with PIL.

    python test_single_image.py ./data/demo_data/messi.jpg --restore_path ./data/darknet_weights/yolov3.weights
"""
from __future__ import division, print_function

import argparse
import sys

import numpy as np

SCORE_THRESH, NMS_THRESH, MAX_BOXES = 0.3, 0.45, 200      # test_single_image.py:57


def parse_args(argv):
    as_bool = lambda text: str(text).lower() == 'true'
    ap = argparse.ArgumentParser(description="YOLO-V3 single image detection.")
    ap.add_argument("input_image", type=str, help="image file to run on")
    ap.add_argument("--anchor_path", type=str, default="./data/yolo_anchors.txt", help="anchor txt file")
    ap.add_argument("--new_size", nargs='*', type=int, default=[416, 416], help="network input size: width height")
    ap.add_argument("--letterbox_resize", type=as_bool, default=True, help="keep the aspect ratio (pad with 128)")
    ap.add_argument("--class_name_path", type=str, default="./data/coco.names", help="class names, one per line")
    ap.add_argument("--restore_path", type=str, default="./data/darknet_weights/yolov3.weights",
                    help="darknet .weights or native .npz checkpoint; random weights if the file does not exist")
    ap.add_argument("--output", type=str, default=None, help="write the annotated image here")
    ap.add_argument("--compute_dtype", type=str, default="f32_wino",
                    help="f32_wino (exact fp32, Winograd 3x3 kernels) | f32 | f32_bf16x6")
    return ap.parse_args(argv)


def to_network_frame(image, size, letterbox):
    """uint8 RGB image -> ([1,h,w,3] float32 in [0,1], function mapping network-frame boxes back to the image)."""
    from yolov3_tensorflow_amd.utils.data_utils import letterbox_resize, _resize
    h0, w0 = image.shape[:2]
    if letterbox:
        resized, ratio, dw, dh = letterbox_resize(image, size[0], size[1])

        def back(boxes):
            boxes[:, [0, 2]] = (boxes[:, [0, 2]] - dw) / ratio
            boxes[:, [1, 3]] = (boxes[:, [1, 3]] - dh) / ratio
            return boxes
    else:
        # cv2.resize(img_ori, tuple(new_size)): OpenCV's default interpolation, INTER_LINEAR (test_single_image.py:43)
        resized = _resize(image, size[0], size[1], 1)

        def back(boxes):
            boxes[:, [0, 2]] *= w0 / float(size[0])
            boxes[:, [1, 3]] *= h0 / float(size[1])
            return boxes
    return (np.asarray(resized, np.float32) / 255.)[np.newaxis], back


def restore(variables, path):
    from yolov3_tensorflow_amd.utils.misc_utils import Saver, load_weights, run_ops
    try:
        if path.endswith('.npz'):
            Saver(variables).restore(path)
        else:
            run_ops(load_weights(variables, path))
        return True
    except (IOError, OSError):
        print('WARNING: %s not found - running with randomly initialised weights' % path)
        return False


def main(argv=None):
    args = parse_args(sys.argv[1:] if argv is None else argv)
    from PIL import Image
    import torch
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.utils.misc_utils import parse_anchors, read_class_names
    from yolov3_tensorflow_amd.utils.plot_utils import get_color_table, plot_one_box

    classes = read_class_names(args.class_name_path)
    model = y3.yolov3(len(classes), parse_anchors(args.anchor_path))
    model.compute_dtype = args.compute_dtype
    picture = np.asarray(Image.open(args.input_image).convert('RGB'))       # RGB; cv2.imread would give BGR
    net_in, back = to_network_frame(picture, args.new_size, args.letterbox_resize)
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3)), False)                   # creates the variables
        restore(y3.global_variables(scope='yolov3'), args.restore_path)
        boxes, scores, labels = model.detect(net_in, max_boxes=MAX_BOXES, score_thresh=SCORE_THRESH,
                                             nms_thresh=NMS_THRESH)[0]
    boxes = back(boxes.cpu().numpy())
    scores, labels = scores.cpu().numpy(), labels.cpu().numpy()

    for title, values in (("box coords:", boxes), ("scores:", scores), ("labels:", labels)):
        print(title)
        print(values)
        print('*' * 30)
    if args.output:
        colours = get_color_table(len(classes))
        canvas = picture.copy()
        for box, score, label in zip(boxes, scores, labels):
            plot_one_box(canvas, box, label=classes[int(label)] + ', {:.2f}%'.format(score * 100),
                         color=colours[int(label)])
        Image.fromarray(canvas).save(args.output)
    return boxes, scores, labels


if __name__ == '__main__':
    main()
