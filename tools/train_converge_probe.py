#!/usr/bin/env python
"""Loss curve of the toy training loop of tests/test_train_script_gpu.py under experiment switches (is a different end
state the chaos of a 200-step Adam memorisation, or a defect?).

    python tools/train_converge_probe.py [--lr-drop]      # env: Y3_WGRAD_OLD_SPLIT=1, Y3_TRAIN_PER_TENSOR_UPDATE=1
"""
import os
import pathlib
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import yolov3_tensorflow_amd as y3
    import train as train_script
    from test_train_script_gpu import make_dataset
    tmp = pathlib.Path(tempfile.mkdtemp())
    ann, names = make_dataset(tmp)
    lr = ['--lr_type', 'fixed']
    if '--lr-drop' in sys.argv:
        lr = ['--lr_type', 'piecewise', '--pw_boundaries', '140', '--pw_values', '1e-3', '1e-4']
    y3.reset_default_graph()
    hist = train_script.main([
        '--train_file', ann, '--val_file', ann, '--restore_path', '', '--save_dir', str(tmp / 'ckpt'),
        '--progress_log_path', '', '--anchor_path', os.path.join(ROOT, 'data', 'yolo_anchors.txt'),
        '--class_name_path', names, '--batch_size', '8', '--img_size', '160', '160', '--letterbox_resize', 'false',
        '--total_epoches', '201', '--train_evaluation_step', '50', '--val_evaluation_epoch', '200',
        '--batch_norm_decay', '0.9', '--save_epoch', '1000', '--optimizer_name', 'adam', '--learning_rate_init', '1e-3'] + lr + [
        '--update_part', 'None', '--multi_scale_train', 'false', '--use_warm_up', 'false', '--warm_up_epoch', '0',
        '--use_label_smooth', 'false', '--use_focal_loss', 'false', '--score_threshold', '0.3', '--nms_topk', '20',
        '--weight_decay', '0'])
    loss = np.array(hist['loss'])
    idx = [0, 1, 2, 3, 5, 10, 20, 50, 100, 120, 140, 150, 160, 170, 180, 190, 200]
    print('PROBE env=%s lr_drop=%s' % ({k: v for k, v in os.environ.items() if k.startswith('Y3_')}, '--lr-drop' in sys.argv))
    print('PROBE loss', ' '.join('%d:%.6g' % (i, loss[i]) for i in idx if i < len(loss)))
    print('PROBE recall', hist['recall'], 'mAP', hist['mAP'])


if __name__ == '__main__':
    main()
