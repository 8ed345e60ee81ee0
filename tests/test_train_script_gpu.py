"""train.py twin end to end (BASELINE configs[3] wiring at toy size): a synthetic 8-image / 3-class set of coloured
rectangles, whole model from random initialisation, Adam.  The step is already pinned tensor by tensor against the
oracle in test_train_gpu.py; this test checks that the LOOP trains: the loss falls from ~1400 to ~1, the model
memorises the set (training-batch recall and mAP > 0.8 through the inference path with the moving BN statistics),
the darknet checkpoint writer runs, and
`restore_include/exclude` + `update_part` select what they say."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_dataset(tmp_path, n=8, size=160, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    colours = [(230, 40, 40), (40, 230, 40), (40, 40, 230)]
    lines = []
    for i in range(n):
        img = np.full((size, size, 3), 110, np.uint8) + rng.randint(0, 20, (size, size, 3)).astype(np.uint8)
        parts = ['%d' % i, str(tmp_path / ('t%d.png' % i)), '%d' % size, '%d' % size]
        for k in range(2):
            w, h = rng.randint(36, 80), rng.randint(36, 80)
            x0, y0 = rng.randint(0, size - w), rng.randint(0, size - h)
            c = (i + k) % 3
            img[y0:y0 + h, x0:x0 + w] = colours[c]
            parts += ['%d' % c, '%d' % x0, '%d' % y0, '%d' % (x0 + w), '%d' % (y0 + h)]
        Image.fromarray(img).save(parts[1])
        lines.append(' '.join(parts))
    ann = tmp_path / 'train.txt'
    ann.write_text('\n'.join(lines) + '\n')
    names = tmp_path / 'names.txt'
    names.write_text('red\ngreen\nblue\n')
    return str(ann), str(names)


def test_training_loop_reduces_the_loss_and_validates(tmp_path, capsys, isolated_graph):
    import yolov3_tensorflow_amd as y3
    sys.path.insert(0, ROOT)
    import train as train_script
    ann, names = make_dataset(tmp_path)
    y3.reset_default_graph()
    hist = train_script.main([
        '--train_file', ann, '--val_file', ann, '--restore_path', '', '--save_dir', str(tmp_path / 'ckpt'),
        '--progress_log_path', str(tmp_path / 'progress.log'), '--anchor_path', os.path.join(ROOT, 'data', 'yolo_anchors.txt'),
        '--class_name_path', names, '--batch_size', '8', '--img_size', '160', '160', '--letterbox_resize', 'false',
        '--total_epoches', '201', '--train_evaluation_step', '50', '--val_evaluation_epoch', '200', '--batch_norm_decay', '0.9', '--save_epoch', '1000',
        '--optimizer_name', 'adam', '--learning_rate_init', '1e-3', '--lr_type', 'piecewise', '--pw_boundaries', '140',
        '--pw_values', '1e-3', '1e-4', '--update_part', 'None',
        '--multi_scale_train', 'false', '--use_warm_up', 'false', '--warm_up_epoch', '0', '--use_label_smooth', 'false',
        '--use_focal_loss', 'false', '--score_threshold', '0.3', '--nms_topk', '20', '--weight_decay', '0', '--augment', 'false',
        '--num_threads', '4'])
    # (learning rate 1e-3 for 140 steps, then 1e-4: at a constant 1e-3 Adam on this 8-image set throws loss spikes
    # in the last 50 steps whose position depends on the last bit of every kernel — tools/train_converge_probe.py
    # shows five numerically equivalent builds agreeing to 1e-5 for the first steps and ending anywhere between
    # loss 0.88 and 2.6; with the drop every one of them settles: loss 1.35, recall 1.0, mAP 1.0)
    out = capsys.readouterr().out
    loss = np.array(hist['loss'])
    print('loss: first %.2f, min %.2f, last %.2f; recalls %s; mAP %s' % (loss[0], loss.min(), loss[-1], hist['recall'],
                                                                       hist['mAP']))
    assert np.isfinite(loss).all()
    assert loss[-10:].mean() < 0.5 * loss[:3].mean()
    assert 'Last batch: rec:' in out and 'EVAL: Recall:' in out
    assert len(hist['mAP']) == 2 and all(0.0 <= m <= 1.0 for m in hist['mAP'])
    # measured: recall on the training batch 1.0 and mAP 1.0 on the (memorised) set after 201 Adam steps (lr drop at 140)
    assert hist['recall'][-1] > 0.8 and hist['mAP'][-1] > 0.8
    files = os.listdir(str(tmp_path / 'ckpt'))
    assert any(f.startswith('best_model_Epoch_') and f.endswith('.weights') for f in files)
    # resume from the native checkpoint (variables + Adam slots + global_step): the loss starts where it ended
    ckpt = sorted(f for f in files if f.endswith('.npz'))[-1]
    y3.reset_default_graph()
    hist2 = train_script.main([
        '--train_file', ann, '--val_file', ann, '--restore_path', str(tmp_path / 'ckpt' / ckpt), '--restore_exclude', 'None',
        '--save_dir', str(tmp_path / 'ckpt2'), '--progress_log_path', '', '--anchor_path',
        os.path.join(ROOT, 'data', 'yolo_anchors.txt'), '--class_name_path', names, '--batch_size', '8',
        '--img_size', '160', '160', '--letterbox_resize', 'false', '--total_epoches', '3', '--train_evaluation_step', '1000',
        '--val_evaluation_epoch', '1000', '--save_epoch', '1000', '--batch_norm_decay', '0.9', '--optimizer_name', 'adam',
        '--learning_rate_init', '1e-3', '--lr_type', 'fixed', '--update_part', 'None', '--multi_scale_train', 'false',
        '--use_warm_up', 'false', '--warm_up_epoch', '0', '--use_label_smooth', 'false', '--use_focal_loss', 'false',
        '--weight_decay', '0', '--augment', 'false', '--num_threads', '4'])
    assert hist2['loss'][0] < 3.0 * loss[-1] and hist2['global_step_start'] == 201
    assert os.path.getsize(str(tmp_path / 'progress.log')) > 0


def test_scope_filters():
    sys.path.insert(0, ROOT)
    import train as train_script

    class V(object):
        def __init__(self, n):
            self.op_name = n
    names = ['yolov3/darknet53_body/Conv/weights', 'yolov3/yolov3_head/Conv_6/weights', 'yolov3/yolov3_head/Conv_6/biases',
             'yolov3/yolov3_head/Conv_60/weights', 'yolov3/yolov3_head/Conv_5/weights']
    vs = [V(n) for n in names]
    sel = lambda inc, exc: [v.op_name for v in train_script.variables_to_restore(vs, inc, exc)]
    assert sel(None, None) == names
    assert sel(None, ['yolov3/yolov3_head/Conv_6']) == [names[0], names[3], names[4]]      # Conv_60 is another scope
    assert sel(['yolov3/yolov3_head'], None) == names[1:]
    assert sel(['yolov3/yolov3_head'], ['yolov3/yolov3_head/Conv_6']) == names[3:]
