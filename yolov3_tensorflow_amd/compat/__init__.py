# coding: utf-8
"""Script-compatibility layer: lets the reference's TF-1 style driver scripts
run BYTE-UNCHANGED on the MI355X-native path, with no TensorFlow and no OpenCV installed.

    python -m yolov3_tensorflow_amd.compat.run <path to the reference's test_single_image.py> ./data/demo_data/messi.jpg \
        --restore_path ./data/darknet_weights/yolov3.ckpt
    python -m yolov3_tensorflow_amd.compat.run <path to the reference's convert_weight.py>
    python -m yolov3_tensorflow_amd.compat.run <path to the reference's eval.py> --eval_file ./data/my_data/val.txt

`run` puts `compat/shims` at the front of sys.path for that one process; the modules there carry the names the scripts
import (ref: test_single_image.py:5-15, convert_weight.py:8-12):

    tensorflow   a deferred graph of exactly the symbols those scripts touch - placeholder, Session.run(fetches,
                 feed_dict), variable_scope, global_variables, train.Saver, the tf.data text-line pipeline + py_func of
                 eval.py - evaluated by this package's eager ops
    cv2          imread / imwrite / resize / cvtColor / rectangle / putText / getTextSize on numpy + PIL; imshow and
                 waitKey do nothing (there is no display)
    model        class yolov3 with the reference's methods, accepting graph tensors
    utils.*      misc_utils, nms_utils, plot_utils, data_aug, data_utils, eval_utils of this package under the
                 reference's module names

It is opt-in and process-local on purpose: a module called `tensorflow` on the default path would shadow a real
TensorFlow.  A checkpoint path is a native one (utils.misc_utils.Saver: `<path>.npz`, keyed by the TF variable names);
a darknet `.weights` file is accepted by `Saver.restore` as well, since TF's own checkpoint format cannot be read
without TensorFlow.
"""
import os
import sys

SHIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shims')


def install():
    """Put the shim modules in front of everything else on sys.path (idempotent)."""
    if sys.path[:1] != [SHIM_DIR]:
        if SHIM_DIR in sys.path:
            sys.path.remove(SHIM_DIR)
        sys.path.insert(0, SHIM_DIR)
    # NumPy-1 spellings the TF-1 era scripts use and NumPy 2 removed (ref: train.py:130 `-np.Inf`)
    import numpy
    for old, new in (('Inf', 'inf'), ('NaN', 'nan'), ('float_', 'float64')):
        if old not in numpy.__dict__:
            setattr(numpy, old, getattr(numpy, new))
    return SHIM_DIR


def dry_run():
    """Y3_COMPAT_DRY_RUN=1: build the graph and walk the script without touching the device (no variables are
    created, Session.run returns each fetch's empty stand-in).  A plumbing check for machines without a GPU."""
    return os.environ.get('Y3_COMPAT_DRY_RUN', '') not in ('', '0')
