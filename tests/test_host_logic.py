"""CPU tests of the host-side logic that needs no device: the darknet loader's traversal, learning-rate
schedules, anchors/class-name parsing, AverageMeter, naming/scoping."""
import os
import tempfile
import types

import numpy as np
import pytest

from yolov3_tensorflow_amd.utils import misc_utils
from yolov3_tensorflow_amd import framework as fw
from oracle import yolo_ref


class FakeVar(object):
    """Stands in for a device variable: records assignments on the host."""

    def __init__(self, name, shape):
        self.name = name + ':0'
        self.shape = fw._Shape(shape)
        self.value = None

    def assign(self, value, validate_shape=True):
        value = np.asarray(value)
        if validate_shape and tuple(value.shape) != tuple(self.shape):
            raise ValueError('shape mismatch for %s' % self.name)
        self.value = value.copy()
        return self

    def numpy(self):
        return self.value


class NamedVar(FakeVar):
    @property
    def op_name(self):
        return self.name[:-2]


def test_load_weights_follows_the_darknet_layout():
    params = yolo_ref.synthetic_params(80, seed=5)
    path = os.path.join(tempfile.mkdtemp(), 'w.weights')
    yolo_ref.write_darknet(params, path)
    var_list = [FakeVar(n, s) for n, s in yolo_ref.variable_specs(80)]
    ops = misc_utils.load_weights(var_list, path)
    assert len(ops) == 366
    misc_utils.run_ops(ops)
    for v in var_list:
        np.testing.assert_array_equal(v.value, params[v.name[:-2]], err_msg=v.name)


def test_save_weights_is_the_inverse_of_load_weights():
    params = yolo_ref.synthetic_params(80, seed=6)
    var_list = [FakeVar(n, s) for n, s in yolo_ref.variable_specs(80)]
    for v in var_list:
        v.assign(params[v.name[:-2]])
    d = tempfile.mkdtemp()
    misc_utils.save_weights(var_list, os.path.join(d, 'a.weights'))
    yolo_ref.write_darknet(params, os.path.join(d, 'b.weights'))
    assert open(os.path.join(d, 'a.weights'), 'rb').read() == open(os.path.join(d, 'b.weights'), 'rb').read()


def test_load_weights_validates_shapes():
    params = yolo_ref.synthetic_params(80, seed=5)
    path = os.path.join(tempfile.mkdtemp(), 'w.weights')
    yolo_ref.write_darknet(params, path)
    specs = yolo_ref.variable_specs(80)
    var_list = [FakeVar(n, s) for n, s in specs]
    ops = misc_utils.load_weights(var_list, path)
    var_list[0].shape = fw._Shape((3, 3, 3, 16))      # variable changed after the ops were built
    with pytest.raises(ValueError):
        ops[4].run()                                  # ops[0..3] are BN params, ops[4] the first kernel


def test_load_weights_restores_a_coco_file_into_another_class_count():
    """fine-tuning on 3 classes from an 80-class darknet file with the default restore_exclude.  The
    loader must cut the stream with the FILE's detection-layer sizes, so that every layer after a detection conv
    (yolov3_head/Conv_7.., Conv_15..) is read from the right offset; the detection convs' ops carry file-shaped arrays
    and are the ones train.py drops (running them raises like tf.assign(validate_shape=True))."""
    params = yolo_ref.synthetic_params(80, seed=5)
    path = os.path.join(tempfile.mkdtemp(), 'coco.weights')
    yolo_ref.write_darknet(params, path)
    var_list = [NamedVar(n, s) for n, s in yolo_ref.variable_specs(3)]
    ops = misc_utils.load_weights(var_list, path)
    exclude = ['yolov3/yolov3_head/Conv_14', 'yolov3/yolov3_head/Conv_6', 'yolov3/yolov3_head/Conv_22']
    keep = set(v.op_name for v in misc_utils.get_variables_to_restore(var_list, None, exclude))
    assert len(keep) == len(var_list) - 6
    misc_utils.run_ops([op for op in ops if op.var.op_name in keep])
    for v in var_list:
        if v.op_name in keep:
            np.testing.assert_array_equal(v.value, params[v.op_name], err_msg=v.op_name)     # incl. the layers AFTER Conv_6
        else:
            assert v.value is None
    det = [op for op in ops if op.var.op_name not in keep]
    assert len(det) == 6
    for op in det:
        with pytest.raises(ValueError):
            op.run()                                   # 255-channel arrays do not fit the 24-channel variables
    # a file that fits no class count is refused
    bad = os.path.join(os.path.dirname(path), 'bad.weights')
    raw = open(path, 'rb').read()
    open(bad, 'wb').write(raw[:-4 * 1000])
    with pytest.raises(ValueError, match='truncated'):
        misc_utils.load_weights(var_list, bad)
    with pytest.raises(ValueError, match='truncated'):
        misc_utils.load_weights([NamedVar(n, s) for n, s in yolo_ref.variable_specs(80)], bad)


def test_parse_anchors_and_class_names(anchors):
    d = tempfile.mkdtemp()
    p = os.path.join(d, 'anchors.txt')
    open(p, 'w').write('10,13, 16,30, 33,23, 30,61, 62,45, 59,119, 116,90, 156,198, 373,326')
    a = misc_utils.parse_anchors(p)
    assert a.dtype == np.float32 and a.shape == (9, 2)
    np.testing.assert_array_equal(a, anchors)
    q = os.path.join(d, 'names.txt')
    open(q, 'w').write('person\nbicycle\ncar\n')
    assert misc_utils.read_class_names(q) == {0: 'person', 1: 'bicycle', 2: 'car'}


def test_average_meter():
    m = misc_utils.AverageMeter()
    m.update(2.0); m.update(4.0, n=3)
    assert m.val == 4.0 and m.count == 4 and m.sum == 14.0 and m.average == 3.5
    m.reset()
    assert m.count == 0 and m.average == 0


def _args(**kw):
    base = dict(learning_rate_init=1e-3, lr_decay_freq=100, lr_decay_factor=0.5, lr_lower_bound=1e-6,
                total_epoches=10, use_warm_up=True, warm_up_epoch=2, train_batch_num=50,
                pw_boundaries=[100, 200], pw_values=[1e-3, 1e-4, 1e-5])
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_learning_rate_schedules():
    lr = misc_utils.config_learning_rate
    assert lr(_args(lr_type='fixed'), 123.0) == 1e-3
    # staircase exponential with lower bound
    assert lr(_args(lr_type='exponential'), 99.0) == pytest.approx(1e-3)
    assert lr(_args(lr_type='exponential'), 100.0) == pytest.approx(5e-4)
    assert lr(_args(lr_type='exponential'), 250.0) == pytest.approx(2.5e-4)
    assert lr(_args(lr_type='exponential', lr_lower_bound=4e-4), 250.0) == pytest.approx(4e-4)
    # piecewise: value i for boundary[i-1] < step <= boundary[i]
    a = _args(lr_type='piecewise')
    assert [lr(a, s) for s in (0.0, 100.0, 101.0, 200.0, 201.0)] == [1e-3, 1e-3, 1e-4, 1e-4, 1e-5]
    # cosine decay: starts at init, ends at the lower bound after train_steps = (10-2)*50
    a = _args(lr_type='cosine_decay')
    assert lr(a, 0.0) == pytest.approx(1e-3)
    assert lr(a, 400.0) == pytest.approx(1e-6)
    assert lr(a, 200.0) == pytest.approx(1e-6 + 0.5 * (1e-3 - 1e-6))
    # cosine restarts: period 100 then 200 (t_mul = 2)
    a = _args(lr_type='cosine_decay_restart')
    assert lr(a, 0.0) == pytest.approx(1e-3)
    assert lr(a, 50.0) == pytest.approx(5e-4)
    assert lr(a, 100.0) == pytest.approx(1e-3)          # restart
    assert lr(a, 200.0) == pytest.approx(5e-4)          # middle of the second (200-step) period
    with pytest.raises(ValueError):
        lr(_args(lr_type='nope'), 0.0)


def test_scope_naming_matches_slim():
    with fw.variable_scope('yolov3'):
        with fw.variable_scope('darknet53_body'):
            names = [fw.unique_layer_name('Conv') for _ in range(3)]
            assert fw.current_scope_name() == 'yolov3/darknet53_body'
        with fw.variable_scope('yolov3_head'):
            again = fw.unique_layer_name('Conv')
    assert names == ['Conv', 'Conv_1', 'Conv_2'] and again == 'Conv'


def test_unknown_optimizer_raises_value_error():
    with pytest.raises(ValueError):
        misc_utils.config_optimizer('lion', 1e-3)


def test_letterbox_resize_matches_the_reference_geometry():
    """messi.jpg is 1296x729 -> ratio 0.32098..., resized 416x234, dw=0, dh=91, pad 128;
    nearest sampling src = min(floor(dst*src/dst_size), src-1)."""
    from yolov3_tensorflow_amd.utils.data_utils import letterbox_resize
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (729, 1296, 3)).astype(np.uint8)
    out, ratio, dw, dh = letterbox_resize(img, 416, 416)
    assert out.shape == (416, 416, 3) and out.dtype == np.uint8
    assert abs(ratio - 416 / 1296) < 1e-12 and (dw, dh) == (0, 91)
    assert (out[:91] == 128).all() and (out[91 + 234:] == 128).all()
    # explicit nearest-neighbour mapping
    sy = min(int(np.floor(10 * 729 / 234.0)), 728)
    sx = min(int(np.floor(100 * 1296 / 416.0)), 1295)
    assert (out[91 + 10, 100] == img[sy, sx]).all()
    # dog.jpg 768x576 -> 416x312, dh=52 ; kite 1352x900 -> 416x276, dh=70
    assert letterbox_resize(np.zeros((576, 768, 3), np.uint8), 416, 416)[2:] == (0, 52)
    assert letterbox_resize(np.zeros((900, 1352, 3), np.uint8), 416, 416)[2:] == (0, 70)
    # interp=1 (cv2.INTER_LINEAR: the eval / validation call sites) shares the geometry
    out1, ratio1, dw1, dh1 = letterbox_resize(img, 416, 416, interp=1)
    assert (ratio1, dw1, dh1) == (ratio, dw, dh) and (out1[:91] == 128).all() and (out1[91 + 234:] == 128).all()
    with pytest.raises(ValueError):
        letterbox_resize(img, 416, 416, interp=2)


def test_bilinear_resize_follows_opencv_inter_linear():
    """utils.data_utils.resize_bilinear_cv2 restates cv2.resize(..., INTER_LINEAR) for uint8 (PIL's BILINEAR
    antialiases on downscale and is a different function).  Parity unpinned (no OpenCV here); checked: half-pixel-centre
    geometry against a float evaluation within 1 LSB, NO antialiasing, border clamping, constant images, the exact 2x
    INTER_AREA path, identity, and known fixed-point answers."""
    from yolov3_tensorflow_amd.utils.data_utils import resize_bilinear_cv2, resize_with_bbox
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    for (nw, nh) in ((416, 416), (20, 11), (53, 80), (106, 74)):
        got = resize_bilinear_cv2(img, nw, nh).astype(np.float64)
        fx = np.clip((np.arange(nw) + 0.5) * (53.0 / nw) - 0.5, 0, 52)
        fy = np.clip((np.arange(nh) + 0.5) * (37.0 / nh) - 0.5, 0, 36)
        x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
        x1, y1 = np.minimum(x0 + 1, 52), np.minimum(y0 + 1, 36)
        ax, ay = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
        v = img.astype(np.float64)
        want = (v[y0][:, x0] * (1 - ax) + v[y0][:, x1] * ax) * (1 - ay) + (v[y1][:, x0] * (1 - ax) + v[y1][:, x1] * ax) * ay
        assert got.shape == (nh, nw, 3) and np.abs(got - want).max() <= 1.0 + 1e-9, (nw, nh, np.abs(got - want).max())
    assert (resize_bilinear_cv2(np.full((9, 7, 3), 201, np.uint8), 31, 17) == 201).all()      # weights sum to 2^11 exactly
    np.testing.assert_array_equal(resize_bilinear_cv2(img, 53, 37), img)                      # identity
    # exact 2x downscale = OpenCV's INTER_AREA fast path: (a+b+c+d+2)>>2
    a = np.array([[10, 20, 30, 41], [50, 60, 70, 80]], np.uint8)
    np.testing.assert_array_equal(resize_bilinear_cv2(a, 2, 1), np.array([[35, 55]], np.uint8))
    # known answers of the fixed-point path: 1x2 -> 1x4 : fx = -0.25, 0.25, 0.75, 1.25 -> [10, 15, 25, 30]
    np.testing.assert_array_equal(resize_bilinear_cv2(np.array([[10, 30]], np.uint8), 4, 1), np.array([[10, 15, 25, 30]], np.uint8))
    # no antialiasing: a 1-pixel checkerboard downscaled 4x samples 2 source pixels per axis, not 16 (PIL would give ~127)
    cb = (np.indices((64, 64)).sum(0) % 2 * 255).astype(np.uint8)
    small = resize_bilinear_cv2(cb, 16, 16)
    assert small.min() == small.max() and 126 <= int(small[0, 0]) <= 129
    # resize_with_bbox (utils/data_aug.py:296-320): plain and letterboxed box mapping
    im, bb = resize_with_bbox(np.zeros((729, 1296, 3), np.uint8), [[100, 50, 300, 250]], 416, 416, interp=1, letterbox=True)
    r = 416 / 1296
    np.testing.assert_allclose(bb, [[100 * r, 50 * r + 91, 300 * r, 250 * r + 91]], rtol=1e-6)
    im, bb = resize_with_bbox(np.zeros((729, 1296, 3), np.uint8), [[100, 50, 300, 250]], 416, 416, interp=1, letterbox=False)
    np.testing.assert_allclose(bb, [[100 / 1296 * 416, 50 / 729 * 416, 300 / 1296 * 416, 250 / 729 * 416]], rtol=1e-6)
    assert im.shape == (416, 416, 3)
    # [N,5] boxes (parse_data's box + mix-up weight, utils/data_utils.py:179-224): the fifth column is carried through
    b5 = np.array([[100, 50, 300, 250, 0.5], [10, 20, 30, 40, 1.0], [0, 0, 8, 8, 0.25], [1, 2, 3, 4, 0.75]], np.float32)
    im, bb = resize_with_bbox(np.zeros((729, 1296, 3), np.uint8), b5.copy(), 416, 416, interp=1, letterbox=False)
    assert bb.shape == (4, 5)
    np.testing.assert_array_equal(bb[:, 4], b5[:, 4])
    np.testing.assert_allclose(bb[0, :4], [100 / 1296 * 416, 50 / 729 * 416, 300 / 1296 * 416, 250 / 729 * 416], rtol=1e-6)
    with pytest.raises(ValueError):
        resize_with_bbox(np.zeros((8, 8, 3), np.uint8), np.zeros((2, 3), np.float32), 4, 4)


def test_native_checkpoint_round_trip_scopes_and_optimizer_slots(tmp_path):
    """Saver = one .npz keyed by the TF variable names; restore honours include/exclude scope
    filters, reports a missing key like TF, and carries the optimizer slots under TF1's slot names."""
    import torch
    from yolov3_tensorflow_amd import training
    params = yolo_ref.synthetic_params(80, seed=7)
    specs = yolo_ref.variable_specs(80)
    src = [NamedVar(n, s) for n, s in specs]
    for v in src:
        v.assign(params[v.op_name])
    opt = training.Optimizer('adam', 1e-3)
    opt.step = 17
    some = src[0]
    opt.slots[some.op_name] = (torch.full(tuple(some.shape), 0.5), torch.full(tuple(some.shape), 0.25))
    path = misc_utils.Saver(src).save(str(tmp_path / 'model-epoch_3'), optimizer=opt, global_step=1234.0)
    assert path.endswith('.npz')
    keys = set(np.load(path).files)
    assert {v.op_name for v in src} <= keys
    assert {some.op_name + '/Adam', some.op_name + '/Adam_1', 'global_step', 'optimizer/step'} <= keys
    # full restore
    dst = [NamedVar(n, s) for n, s in specs]
    assert misc_utils.Saver(dst).restore(path) == 1234.0
    for v in dst:
        np.testing.assert_array_equal(v.value, params[v.op_name])
    # the reference's fine-tuning recipe: everything except the three detection convs (args.py:52-55)
    exclude = ['yolov3/yolov3_head/Conv_14', 'yolov3/yolov3_head/Conv_6', 'yolov3/yolov3_head/Conv_22']
    dst = [NamedVar(n, s) for n, s in specs]
    part = misc_utils.get_variables_to_restore(dst, None, exclude)
    assert len(part) == len(dst) - 6           # three convs x (weights, biases)
    misc_utils.Saver(part).restore(path)
    assert all((v.value is None) == any(v.op_name.startswith(e + '/') for e in exclude) for v in dst)
    body = misc_utils.get_variables_to_restore(dst, ['yolov3/darknet53_body'], None)
    assert len(body) == 52 * 5 and all(v.op_name.startswith('yolov3/darknet53_body/') for v in body)
    # a variable the checkpoint does not hold
    with pytest.raises(KeyError):
        misc_utils.Saver([NamedVar('yolov3/yolov3_head/Conv_99/weights', (1, 1, 4, 4))]).restore(path)
    # shape mismatch is caught by assign(validate_shape=True)
    with pytest.raises(ValueError):
        misc_utils.Saver([NamedVar(specs[0][0], (1, 1, 1, 1))]).restore(path)
    # optimizer slots come back under the same names
    opt2 = training.Optimizer('adam', 1e-3)

    class DevVar(NamedVar):
        tensor = torch.zeros(1)
    dv = DevVar(specs[0][0], specs[0][1])
    misc_utils.Saver([dv]).restore(path, optimizer=opt2)
    assert opt2.step == 17 and float(opt2.slots[dv.op_name][0].flatten()[0]) == 0.5 and \
        float(opt2.slots[dv.op_name][1].flatten()[0]) == 0.25
    # slots + step only — the variable itself is left alone, a slot of another shape is skipped
    opt3 = training.Optimizer('adam', 1e-3)
    dv3 = DevVar(specs[0][0], specs[0][1])
    misc_utils.Saver([dv3]).restore(path, optimizer=opt3, variables=False)
    assert dv3.value is None and opt3.step == 17 and dv3.op_name in opt3.slots
    opt4 = training.Optimizer('adam', 1e-3)
    other = DevVar(specs[0][0], (1, 1, 1, 1))
    misc_utils.Saver([other]).restore(path, optimizer=opt4, variables=False)
    assert other.value is None and other.op_name not in opt4.slots



def test_bench_reports_traffic_only_from_a_profile_of_this_build(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` comes from a committed PMC pass (rocprofv3 cannot run inside the bench process) and
    is reported ONLY when that file is stamped with the hash of this build's kernel sources; a file taken on other
    sources gives traffic = None and a source line that says why (no stale bytes next to fresh timings)."""
    import importlib.util
    import json
    from yolov3_tensorflow_amd.build import csrc_sha16
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # the committed round-2 file: parses, self-consistent, unstamped -> stale on any later build
    with open(os.path.join(root, 'profiles', 'r02_pmc_traffic_wino.json')) as f:
        d = json.load(f)
    assert d['read_bytes_per_launch'] + d['write_bytes_per_launch'] == d['traffic_bytes_per_launch']
    assert abs(d['calibration']['read_factor'] - 1.0) < 1e-3 and abs(d['calibration']['write_factor'] - 1.0) < 1e-3
    assert len(d['per_layer']) == 75 and sum('conv_wino' in r['kernel'] for r in d['per_layer']) == 32
    if d.get('csrc_sha16') != csrc_sha16():
        val, src = bench.traffic_from_profile(['r02_pmc_traffic_wino.json'])
        assert val is None and 'not reported' in src
    # a file stamped with this build's hash is reported; one with another hash is not
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'profiles')
    for name, sha in (('fresh.json', csrc_sha16()), ('old.json', '0' * 16)):
        with open(tmp_path / 'profiles' / name, 'w') as f:
            json.dump({'traffic_bytes_per_launch': 123456789, 'csrc_sha16': sha}, f)
    val, src = bench.traffic_from_profile(['old.json', 'fresh.json'])
    assert val == 123456789 and 'fresh.json' in src
    val, src = bench.traffic_from_profile(['old.json'])
    assert val is None and 'old.json' in src
    assert bench.traffic_from_profile(['does_not_exist.json']) == (None, None)
