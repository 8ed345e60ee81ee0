"""oracle/ — CPU restatement of the reference's algorithm for the YOLOv3 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under yolov3_tensorflow_amd/ imports this package; only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may use it, and only as the checker.

Parity status: the reference ships no tests, golden tensors or known-answer vectors,
and its arithmetic executes inside TensorFlow 1.x (unpinned, not installed here, no network).  Hence:
  * numpy-only reference functions (py_nms, cpu_nms, process_box, parse_anchors) ARE pinned: they are
    imported unmodified from /root/reference under stub `tensorflow`/`cv2` modules by
    tests/golden/make_golden.py, and their outputs are committed under tests/golden/;
  * every TensorFlow-defined op (conv/BN/leaky, resize_nearest, sigmoid/exp decode,
    tf.image.non_max_suppression, the loss graph, optimizers) is restated from the reference's call
    sites and TF's documented semantics: **parity unpinned** for those (no TF to run against).
"""
