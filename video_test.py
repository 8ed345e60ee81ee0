# coding: utf-8
"""Detect the objects of every frame of a video on the MI355X-native path.

Command line of the reference's video_test.py (positional video path, --anchor_path, --new_size, --letterbox_resize,
--class_name_path, --restore_path, --save_video).  What the environment changes: there is no OpenCV / ffmpeg here, so the
video is Motion-JPEG in an AVI container (or an animated image, or a directory of frames: utils.video_utils), the result
goes to --output (default video_result.avi) instead of video_result.mp4, and instead of an OpenCV window the per-frame
detections are counted on stdout; weights come from a darknet `.weights` file or a native `.npz` checkpoint.  What the
device path changes: frames go through forward -> decode -> per-class NMS in batches of --batch_size (the reference runs
one frame per sess.run), and the time drawn on each frame is the batch's time divided by its frames.

    python video_test.py ./data/demo_data/video.avi --restore_path ./data/darknet_weights/yolov3.weights --save_video true
"""
from __future__ import division, print_function

import argparse
import sys
import time

import numpy as np

from test_single_image import MAX_BOXES, NMS_THRESH, SCORE_THRESH, restore, to_network_frame      # video_test.py:63


def parse_args(argv):
    as_bool = lambda text: str(text).lower() == 'true'
    ap = argparse.ArgumentParser(description="YOLO-V3 video test procedure.")
    ap.add_argument("input_video", type=str, help="Motion-JPEG AVI, animated image or directory of frames")
    ap.add_argument("--anchor_path", type=str, default="./data/yolo_anchors.txt", help="anchor txt file")
    ap.add_argument("--new_size", nargs='*', type=int, default=[416, 416], help="network input size: width height")
    ap.add_argument("--letterbox_resize", type=as_bool, default=True, help="keep the aspect ratio (pad with 128)")
    ap.add_argument("--class_name_path", type=str, default="./data/coco.names", help="class names, one per line")
    ap.add_argument("--restore_path", type=str, default="./data/darknet_weights/yolov3.weights",
                    help="darknet .weights or native .npz checkpoint; random weights if the file does not exist")
    ap.add_argument("--save_video", type=as_bool, default=False, help="write the annotated frames to --output")
    ap.add_argument("--output", type=str, default="video_result.avi", help="annotated video (Motion-JPEG AVI)")
    ap.add_argument("--batch_size", type=int, default=8, help="frames per device batch (the reference runs 1)")
    ap.add_argument("--compute_dtype", type=str, default="f32_wino",
                    help="f32_wino (exact fp32, Winograd 3x3 kernels) | f32 | f32_bf16x6")
    return ap.parse_args(argv)


def annotate(frame, boxes, scores, labels, classes, colours, ms):
    """Boxes with 'class, score%' captions and the frame time in the top-left corner (video_test.py:98-103), in place."""
    from PIL import Image, ImageDraw
    from yolov3_tensorflow_amd.utils.plot_utils import plot_one_box
    for box, score, label in zip(boxes, scores, labels):
        plot_one_box(frame, box, label=classes[int(label)] + ', {:.2f}%'.format(score * 100), color=colours[int(label)])
    canvas = Image.fromarray(frame)
    ImageDraw.Draw(canvas).text((40, 20), '{:.2f}ms'.format(ms), fill=(0, 255, 0))
    frame[...] = np.asarray(canvas)
    return frame


def main(argv=None):
    args = parse_args(sys.argv[1:] if argv is None else argv)
    import torch
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.utils.misc_utils import parse_anchors, read_class_names
    from yolov3_tensorflow_amd.utils.plot_utils import get_color_table
    from yolov3_tensorflow_amd.utils.video_utils import MjpegAviWriter, open_video

    classes = read_class_names(args.class_name_path)
    colours = get_color_table(len(classes))
    video = open_video(args.input_video)
    print('%s: %d frames, %dx%d, %.2f fps' % (args.input_video, video.frame_count, video.width, video.height, video.fps))
    writer = MjpegAviWriter(args.output, video.fps or 25.0, (video.width, video.height)) if args.save_video else None

    model = y3.yolov3(len(classes), parse_anchors(args.anchor_path))
    model.compute_dtype = args.compute_dtype
    results = []
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3)), False)                   # creates the variables
        restore(y3.global_variables(scope='yolov3'), args.restore_path)
        done = False
        while not done:
            frames, inputs, backs = [], [], []
            while len(frames) < max(1, args.batch_size):
                frame = video.read()
                if frame is None:
                    done = True
                    break
                net_in, back = to_network_frame(frame, args.new_size, args.letterbox_resize)
                frames.append(np.array(frame))
                inputs.append(net_in[0])
                backs.append(back)
            if not frames:
                break
            start = time.time()
            dets = model.detect(np.stack(inputs), max_boxes=MAX_BOXES, score_thresh=SCORE_THRESH, nms_thresh=NMS_THRESH)
            dets = [tuple(t.cpu().numpy() for t in det) for det in dets]      # (the copy waits for the device)
            ms = (time.time() - start) * 1000 / len(frames)
            for frame, back, (boxes, scores, labels) in zip(frames, backs, dets):
                boxes = back(boxes)
                results.append((boxes, scores, labels))
                print('frame %d: %d detections, %.2f ms' % (len(results) - 1, len(boxes), ms))
                if writer is not None:
                    writer.write(annotate(frame, boxes, scores, labels, classes, colours, ms))
    video.close()
    if writer is not None:
        writer.close()
        print('annotated video written to %s' % args.output)
    return results


if __name__ == '__main__':
    main()
