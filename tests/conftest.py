import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'reference_numpy_goldens.npz')

# the 9 COCO anchors of the reference's data/yolo_anchors.txt (also pinned in the golden file)
COCO_ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326],
                        np.float32).reshape(9, 2)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a box without a HIP device SKIPS the gpu-marked tests instead of failing them
    (ADVICE r2); `-m gpu` on the GPU box and `-m "not gpu"` on the CPU box are unaffected."""
    gpu_items = [it for it in items if it.get_closest_marker('gpu') is not None]
    if not gpu_items:
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device visible here)")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    return np.load(GOLDEN)


@pytest.fixture(scope='session')
def anchors():
    return COCO_ANCHORS.copy()


def make_boxes(rng, n, size=416.0, wmin=8.0, wmax=256.0):
    c = rng.uniform(0, size, (n, 2))
    wh = rng.uniform(wmin, wmax, (n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)


def blob_images(seed, n, hw):
    """Structured synthetic images in [0,1): sums of block-upsampled random grids, so activations vary in
    space (white noise gives spatially uniform statistics)."""
    rng = np.random.RandomState(seed)
    img = np.zeros((n, hw, hw, 3), np.float32)
    for g, wt in ((hw // 32, 0.45), (hw // 8, 0.35), (hw, 0.2)):
        t = rng.rand(n, g, g, 3).astype(np.float32)
        img += wt * np.repeat(np.repeat(t, hw // g, 1), hw // g, 2)
    return img


@pytest.fixture(scope='session')
def gpu_model():
    """A yolov3 with the oracle's synthetic weights loaded THROUGH the darknet-format loader
    (oracle writes the file, the product's load_weights reads it)."""
    import tempfile
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd.utils import misc_utils
    from oracle import yolo_ref
    params = yolo_ref.synthetic_params(80, seed=1)
    path = os.path.join(tempfile.mkdtemp(), 'synthetic.weights')
    yolo_ref.write_darknet(params, path)
    y3.reset_default_graph()
    model = y3.yolov3(80, COCO_ANCHORS)
    import torch
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 64, 64, 3))       # creates the variables (like building the graph)
    ops = misc_utils.load_weights(y3.global_variables(scope='yolov3'), path)
    misc_utils.run_ops(ops)
    return model, params


@pytest.fixture
def isolated_graph():
    """Run a test on an empty variable store and put the previous one back afterwards, so that tests which build
    their own networks (y3.reset_default_graph(), the script twins) cannot invalidate the session-scoped `gpu_model`."""
    from yolov3_tensorflow_amd import framework as fw
    saved = dict(fw._VARIABLES)
    fw._VARIABLES.clear()
    fw._bump_global_version()
    try:
        yield
    finally:
        fw._VARIABLES.clear()
        fw._VARIABLES.update(saved)
        fw._bump_global_version()
