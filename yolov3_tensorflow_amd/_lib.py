"""ctypes binding of libyolo355.so (include/yolo355.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every op raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("Y3_LIB_PATH") or os.path.join(HERE, "csrc", "libyolo355.so")   # (override: probe builds)

Y3_OK, Y3_EINVAL, Y3_EHIP, Y3_ESTATE = 0, -1, -2, -3
Y3_NMS_TF, Y3_NMS_PY = 0, 1


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("n", "h", "w", "cin", "c_up", "cout", "k", "stride", "act")]


class ParamDesc(ctypes.Structure):      # y3_param_desc
    _fields_ = [("w", c_void_p), ("g", c_void_p), ("slot0", c_void_p), ("slot1", c_void_p), ("n", c_longlong),
                ("weight_decay", c_float), ("reserved", c_int)]


class TrainVar(ctypes.Structure):       # y3_train_var
    _fields_ = [("weights", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("moving_mean", c_void_p),
                ("moving_variance", c_void_p), ("biases", c_void_p), ("g_weights", c_longlong), ("g_gamma", c_longlong),
                ("g_beta", c_longlong), ("g_biases", c_longlong), ("g_end", c_longlong)]


class TrainOpts(ctypes.Structure):      # y3_train_opts
    _fields_ = [("bn_decay", c_float), ("use_label_smooth", c_int), ("use_focal_loss", c_int),
                ("anchors", POINTER(c_float))]


GradReadyFn = ctypes.CFUNCTYPE(None, c_void_p, c_longlong)      # y3_grad_ready_fn


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header.
PROTOTYPES = {
    "y3_last_error": (c_char_p, []),
    "y3_abi_version": (c_int, []),
    "y3_ctx_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "y3_ctx_destroy": (c_int, [c_void_p]),
    "y3_ctx_check": (c_int, [c_void_p]),
    "y3_debug_streamk_fault": (None, [c_int]),
    "y3_pack_conv_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "y3_bn_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p]),
    "y3_conv_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "y3_streamk_range": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, POINTER(ctypes.c_longlong),
                                 POINTER(ctypes.c_longlong)]),
    "y3_conv2d_fwd": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_size_t]),
    "y3_conv_wino_eligible": (c_int, [POINTER(ConvDesc)]),
    "y3_pack_conv_weights_wino": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "y3_conv_wino_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "y3_conv_wino44_eligible": (c_int, [POINTER(ConvDesc)]),
    "y3_conv_wino44_candidate": (c_int, [POINTER(ConvDesc)]),
    "y3_conv_wino44_preferred": (c_int, [POINTER(ConvDesc)]),
    "y3_pack_conv_weights_wino44": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "y3_conv_wino44_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "y3_conv2d_fwd_wino44": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t]),
    "y3_conv2d_fwd_wino44_stats": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_size_t]),
    "y3_pack_conv_weights_wino44_dgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "y3_conv2d_dgrad_wino44": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                       c_void_p, c_size_t]),
    "y3_conv2d_fwd_wino": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t]),
    "y3_pack_conv_weights_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "y3_conv2d_fwd_split": (c_int, [c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "y3_pack_conv_weights_split_dgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "y3_conv2d_dgrad_split": (c_int, [c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      c_int, c_void_p, c_void_p, c_size_t]),
    "y3_pack_conv_weights_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "y3_conv_bf16_tile": (c_int, [POINTER(ConvDesc)]),
    "y3_conv2d_fwd_bf16": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int]),
    "y3_conv2d_fwd_bf16_stem_s2": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "y3_conv2d_fwd_stem_s2": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "y3_resblock64_fwd_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "y3_net_train_workspace_bytes": (c_size_t, [c_void_p, POINTER(TrainVar), c_int, c_int, c_int]),
    "y3_net_train_forward": (c_int, [c_void_p, POINTER(TrainVar), c_void_p, c_int, c_int, c_int, POINTER(TrainOpts),
                                     c_void_p, c_size_t, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "y3_net_train_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(TrainOpts), c_void_p]),
    "y3_net_train_backward": (c_int, [c_void_p, POINTER(TrainVar), c_void_p, GradReadyFn, c_void_p]),
    "y3_net_train_step": (c_int, [c_void_p, POINTER(TrainVar), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  POINTER(TrainOpts), c_void_p, c_void_p, c_size_t, c_void_p, GradReadyFn, c_void_p]),
    "y3_net_train_set_wgrad_stream": (c_int, [c_void_p, c_void_p]),
    "y3_net_train_saved": (c_int, [c_void_p, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "y3_net_set_dtype": (c_int, [c_void_p, c_int]),
    "y3_upsample_nearest": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "y3_concat_channels": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_longlong, c_void_p]),
    "y3_add": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    "y3_reorg_boxes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                               POINTER(c_float), c_void_p]),
    "y3_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                          POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p]),
    "y3_nms_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "y3_nms": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                       c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "y3_net_create": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "y3_net_destroy": (c_int, [c_void_p]),
    "y3_net_num_layers": (c_int, [c_void_p]),
    "y3_net_layer_info": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                  POINTER(c_int), POINTER(c_int)]),
    "y3_net_set_layer": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "y3_net_set_layer_alt": (c_int, [c_void_p, c_int, c_void_p]),
    "y3_net_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "y3_net_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                               c_void_p, c_void_p]),
    "y3_net_layer_graph": (c_int, [c_void_p, c_int] + [POINTER(c_int)] * 5),
    "y3_net_num_tensors": (c_int, [c_void_p]),
    "y3_net_tensor_info": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "y3_reduce_scratch_bytes": (c_size_t, [c_int]),
    "y3_bn_train_stats": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_float,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "y3_conv_stats_blocks": (c_int, [POINTER(ConvDesc), c_int]),
    "y3_conv2d_fwd_stats": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t]),
    "y3_conv2d_fwd_wino_stats": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_size_t]),
    "y3_bn_train_stats_partials": (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_void_p, c_void_p, c_float,
                                           c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "y3_bn_apply_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    "y3_bn_bwd_scratch_bytes": (c_size_t, [c_int]),
    "y3_bn_train_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "y3_bias_grad": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p]),
    "y3_conv2d_dgrad": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_void_p, c_size_t]),
    "y3_pack_conv_weights_wino_dgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "y3_conv2d_dgrad_wino": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_size_t]),
    "y3_conv_wgrad_scratch_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "y3_conv_wgrad": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t]),
    "y3_conv_wgrad_wino_eligible": (c_int, [POINTER(ConvDesc)]),
    "y3_conv_wgrad_wino_scratch_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "y3_conv_wgrad_wino": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t]),
    "y3_upsample2x_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "y3_slice_accumulate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_int, c_void_p]),
    "y3_pad_channels": (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_void_p]),
    "y3_loss_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "y3_loss_layer": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                              POINTER(c_float), c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t]),
    "y3_process_box": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               POINTER(c_float), c_void_p, c_void_p, c_void_p]),
    "y3_feed_run": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int]),
    "y3_box_iou": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_int, c_void_p]),
    "y3_optimizer_scratch_bytes": (c_size_t, []),
    "y3_clip_update": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float,
                               c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p]),
    "y3_clip_update_multi_scratch_bytes": (c_size_t, [c_void_p, c_int]),
    "y3_clip_update_multi": (c_int, [c_void_p, c_int, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float,
                                     c_float, c_float, c_void_p, c_size_t]),
    "y3_net_set_profiling": (c_int, [c_void_p, c_int]),
    "y3_net_get_layer_ms": (c_int, [c_void_p, POINTER(c_float), c_int]),
    "y3_net_layer_is_streamk": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "y3_net_layer_fused": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
}

_lib = None


class Y3Error(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Y3Error(
                "libyolo355.so is not built (%s). Run `python -m yolov3_tensorflow_amd.build` "
                "(needs hipcc). There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.y3_abi_version() != 3:
            raise Y3Error("libyolo355.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc):
    """Map a C status to the exception types the reference's callers see."""
    if rc == Y3_OK:
        return
    msg = lib().y3_last_error().decode(errors="replace")
    if rc == Y3_EINVAL:
        raise ValueError(msg)
    raise Y3Error("%s (status %d)" % (msg, rc))
