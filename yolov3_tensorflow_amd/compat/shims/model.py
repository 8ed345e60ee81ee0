# coding: utf-8
"""`model` as the reference's scripts import it (ref: model.py:14-30,140; `from model import yolov3`): this package's
yolov3 whose methods also accept the graph tensors of the `tensorflow` shim.  With a graph input, forward() creates the
366 variables right away (so that tf.train.Saver / tf.global_variables see them before the first Session.run, as in
TF) and defers the computation to Session.run."""
import numpy as _np

import yolov3_tensorflow_amd as _y3
from yolov3_tensorflow_amd import compat as _compat, framework as _fw
from yolov3_tensorflow_amd.compat import lazy as _lazy


class yolov3(_y3.yolov3):

    def __init__(self, *args, **kwargs):
        _y3.yolov3.__init__(self, *args, **kwargs)
        import tensorflow as _tf            # (the shim: tf.losses.get_regularization_loss reads the model's weight decay)
        _tf._REG['weight_decay'] = float(self.weight_decay)

    def forward(self, inputs, is_training=False, reuse=False):
        if not _lazy.is_node(inputs):
            return _y3.yolov3.forward(self, inputs, is_training, reuse)
        scope = _fw.current_scope_name()
        shape = inputs.get_shape()
        if shape is not None and len(shape) == 4 and shape[1] and shape[2]:
            self.img_size = [int(shape[1]), int(shape[2])]
        if not _compat.dry_run():
            with _y3.variable_scope_absolute(scope):
                self._get_net(_fw.default_device())            # creates the variables in the reference's order

        def run(x, training_flag):
            # eval.py feeds its `is_training` placeholder with False (ref: eval.py:66,116), train.py with True for the
            # training batches and False for validation (ref: train.py:140,184)
            self.img_size = [int(x.shape[1]), int(x.shape[2])]
            with _y3.variable_scope_absolute(scope):
                return _y3.yolov3.forward(self, x, bool(training_flag), reuse)

        det = 3 * (5 + self.class_num)
        empties = [_np.zeros((0, 0, 0, det), _np.float32)] * 3
        return _lazy.multi(run, (inputs, is_training), 3, 'yolov3/forward', empties)

    def compute_loss(self, y_pred, y_true):
        if not (_lazy.is_node(y_pred) or _lazy.is_node(y_true)):
            return _y3.yolov3.compute_loss(self, y_pred, y_true)

        def run(*args):
            loss = _y3.yolov3.compute_loss(self, list(args[:3]), list(args[3:]))
            return tuple(float(v) for v in loss)

        outs = list(_lazy.multi(run, tuple(y_pred) + tuple(y_true), 5, 'yolov3/compute_loss', [_np.float32(0.0)] * 5))
        for o in outs:
            o.model = self              # (the compat optimizer finds the model to train through the loss tensor)
        return outs

    def predict(self, feature_maps, with_scores=False):
        if not _lazy.is_node(feature_maps):
            return _y3.yolov3.predict(self, feature_maps, with_scores)
        n = 4 if with_scores else 3
        c = self.class_num
        empties = [_np.zeros((1, 0, 4), _np.float32), _np.zeros((1, 0, 1), _np.float32),
                   _np.zeros((1, 0, c), _np.float32), _np.zeros((1, 0, c), _np.float32)][:n]
        return _lazy.multi(lambda *fms: _y3.yolov3.predict(self, list(fms), with_scores), tuple(feature_maps), n,
                           'yolov3/predict', empties)
