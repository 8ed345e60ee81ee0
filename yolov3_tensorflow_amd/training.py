"""Training path (train.py:72-115 of the reference): BN in batch-statistics mode, loss, backward, clip,
optimizer update.  Optimizer is the object utils.misc_utils.config_optimizer returns."""


class Optimizer(object):
    """Hyper-parameters of one of the four update rules the reference can select
    (utils/misc_utils.py:151-161; TF1 definitions)."""

    KINDS = ('sgd', 'momentum', 'adam', 'rmsprop')

    def __init__(self, kind, learning_rate, momentum=0.9, decay=0.9, beta1=0.9, beta2=0.999,
                 epsilon=None):
        if kind not in self.KINDS:
            raise ValueError('Unsupported optimizer type!')
        self.kind = kind
        self.learning_rate = learning_rate   # float, or a callable step -> float
        self.momentum = momentum
        self.decay = decay
        self.beta1, self.beta2 = beta1, beta2
        self.epsilon = epsilon if epsilon is not None else (1e-8 if kind == 'adam' else 1e-10)
        self.slots = {}                      # variable op_name -> tuple of device tensors
        self.step = 0

    def lr_at(self, global_step):
        return self.learning_rate(global_step) if callable(self.learning_rate) else float(self.learning_rate)


def forward_train(model, x):
    raise NotImplementedError('yolov3.forward(is_training=True): the training kernels are not built yet')


def loss_layer(model, feature_map_i, y_true, anchors):
    raise NotImplementedError('loss_layer: the training kernels are not built yet')


def box_iou(pred_boxes, valid_true_boxes):
    raise NotImplementedError('box_iou: the training kernels are not built yet')


def compute_loss(model, y_pred, y_true):
    raise NotImplementedError('compute_loss: the training kernels are not built yet')
