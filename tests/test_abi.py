"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/yolo355.h declares, and reports errors through the status/last_error channel.  No compute."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'yolo355.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(y3_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    from yolov3_tensorflow_amd import build, _lib
    build.build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    return _lib.lib()


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ('y3_conv2d_fwd', 'y3_decode', 'y3_nms', 'y3_net_forward', 'y3_pack_conv_weights',
                 'y3_bn_fold', 'y3_last_error', 'y3_ctx_create'):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    from yolov3_tensorflow_amd import _lib
    for name in declared_symbols():
        assert hasattr(lib, name), 'libyolo355.so does not export %s' % name
        assert name in _lib.PROTOTYPES, 'no ctypes prototype for %s' % name
    assert sorted(_lib.PROTOTYPES) == declared_symbols()


def test_abi_version_and_pure_host_entry_points(lib):
    assert lib.y3_abi_version() == 3
    # workspace sizing is host-only arithmetic
    assert lib.y3_nms_workspace_bytes(1, 10647, 80, 200) > 80 * 10647 * 4
    assert lib.y3_nms_workspace_bytes(0, 10647, 80, 200) == 0
    assert lib.y3_net_num_layers(None) == 0


def test_errors_are_reported_not_thrown(lib):
    import torch
    from yolov3_tensorflow_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('error path for a missing GPU is only observable without one')
    out = ctypes.c_void_p()
    rc = lib.y3_ctx_create(0, None, ctypes.byref(out))
    assert rc in (_lib.Y3_EHIP, _lib.Y3_EINVAL)
    assert len(lib.y3_last_error()) > 0
    with pytest.raises((_lib.Y3Error, ValueError)):
        _lib.check(rc)
    assert lib.y3_ctx_create(0, None, None) == _lib.Y3_EINVAL


def test_product_fails_loudly_without_gpu():
    import torch
    import numpy as np
    if torch.cuda.is_available():
        pytest.skip('needs a GPU-less host')
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import _lib
    from conftest import COCO_ANCHORS
    model = y3.yolov3(80, COCO_ANCHORS)
    with pytest.raises(_lib.Y3Error):
        model.forward(np.zeros((1, 64, 64, 3), np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'yolov3_tensorflow_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'oracle/' not in src or f.endswith('.md'), f


def test_net_graph_matches_oracle_variable_specs_and_survey_totals(lib):
    """Host-only: the C++ launch plan (y3_net) and the oracle's traced graph are independent derivations of
    the same 75-conv network; they must agree layer by layer, and reproduce SURVEY.md's totals."""
    import numpy as np
    from oracle import yolo_ref
    sys_path_bench = __import__('importlib').import_module('bench')
    h = ctypes.c_void_p()
    assert lib.y3_net_create(None, 80, ctypes.byref(h)) == 0
    n = lib.y3_net_num_layers(h)
    assert n == 75
    table = []
    for i in range(n):
        v = [ctypes.c_int() for _ in range(5)]
        assert lib.y3_net_layer_info(h, i, *[ctypes.byref(x) for x in v]) == 0
        table.append(tuple(x.value for x in v))
    kernels = [(name, shape) for name, shape in yolo_ref.variable_specs(80) if name.endswith('/weights')]
    assert len(kernels) == 75
    for (k, s, cin, cout, bn), (name, shape) in zip(table, kernels):
        assert shape == (k, k, cin, cout), name
    assert sum(1 for t in table if t[4]) == 72 and sum(1 for t in table if t[1] == 2) == 5
    flops = sys_path_bench.conv_flops(table, 1, 416, 416)
    assert abs(flops.sum() / 1e9 - 65.864) < 0.01
    assert abs(sys_path_bench.conv_flops(table, 1, 608, 608).sum() / 1e9 - 140.692) < 0.01
    # workspace plan: liveness reuse keeps the arena far below the sum of all activations
    ws = lib.y3_net_workspace_bytes(h, 32, 416, 416)
    total_act = 39.2e6 * 4 * 32
    assert 708.8e6 + 354e6 <= ws < 0.4 * total_act
    assert lib.y3_net_workspace_bytes(h, 32, 400, 416) == 0  # not a multiple of 32
    # forward without a context / parameters is refused, not crashed
    from yolov3_tensorflow_amd import _lib as L
    dummy = ctypes.c_void_p(256)
    assert lib.y3_net_forward(h, dummy, 1, 64, 64, dummy, ctypes.c_size_t(1 << 30), dummy, dummy, dummy) == L.Y3_ESTATE
    assert lib.y3_net_destroy(h) == 0
