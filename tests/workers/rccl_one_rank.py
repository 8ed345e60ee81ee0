"""Worker of tests/test_rccl_gpu.py, launched under `python -m torch.distributed.run --nproc-per-node 1`: one train step
through training.Trainer(process_group=WORLD) on the `nccl` (= RCCL) backend with the collectives FORCED in the one-rank
group, written out for the parent to compare with a no-group step."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from conftest import COCO_ANCHORS, blob_images      # noqa: E402

N, SIZE, LR, FACTOR = 2, 96, 1e-3, 2.0


def setup():
    import yolov3_tensorflow_amd as y3
    from oracle import yolo_ref
    params = yolo_ref.synthetic_params(80, seed=3)
    y3.reset_default_graph()
    model = y3.yolov3(80, COCO_ANCHORS, batch_norm_decay=0.9, weight_decay=5e-4)
    model.compute_dtype = 'f32_wino'
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros(1, 32, 32, 3))
    for v in y3.global_variables(scope='yolov3'):
        v.assign(params[v.op_name])
    return y3, model


def data():
    from oracle import train_ref
    return blob_images(40, N, SIZE), train_ref.synthetic_targets(50, N, [SIZE, SIZE], 80, COCO_ANCHORS, max_boxes=3)


def one_step(group, premul):
    """Returns (variables after the step, flat gradient buffer, meta)."""
    from yolov3_tensorflow_amd import training
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    y3, model = setup()
    trainer = training.Trainer(model, config_optimizer('momentum', LR), process_group=group, bucket_bytes=16 << 20)
    x, yts = data()
    meta = {}
    with y3.variable_scope('yolov3'):
        fms = model.forward(x, is_training=True)
        training.compute_loss(model, fms, yts)
        if group is not None:
            trainer._alloc_grads(model._train['layer_vars'], fms[0].device)
            trainer.exchange.force = True
            if premul:
                trainer.exchange.op = dist._make_nccl_premul_sum(FACTOR)
        elif premul:
            trainer._alloc_grads(model._train['layer_vars'], fms[0].device)
            trainer.exchange.world = 1.0 / FACTOR          # no group: the factor goes through grad_scale = 1 / world
        trainer.backward()
        meta['issued_before_end'] = len(trainer.exchange.issued)
        meta['buckets'] = len(trainer.exchange.edges)
        meta['works_in_flight'] = len(trainer.exchange._works)
        meta['compute_stream'] = int(torch.cuda.current_stream().cuda_stream)
        trainer.apply_gradients()
    torch.cuda.synchronize()
    return ({v.op_name: v.numpy() for v in y3.global_variables(scope='yolov3')}, trainer.flat.cpu().numpy(), meta)


def main():
    out_dir = sys.argv[1]
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']),
                            device_id=torch.device('cuda', local))
    meta = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
    try:
        dist._make_nccl_premul_sum(FACTOR)
        premul = True
    except Exception as e:      # noqa: BLE001 - recorded, the parent decides
        premul = False
        meta['premul_error'] = '%s: %s' % (type(e).__name__, e)
    meta['premul'] = premul
    for tag, pm in (('sum', False),) + ((('premul', True),) if premul else ()):
        try:
            v_g, g_g, m_g = one_step(dist.group.WORLD, pm)
        except Exception as e:  # noqa: BLE001
            if pm:              # the backend may accept the op object and reject it at launch
                meta['premul'] = False
                meta['premul_error'] = '%s: %s' % (type(e).__name__, e)
                continue
            raise
        v_l, g_l, m_l = one_step(None, pm)
        np.savez(os.path.join(out_dir, '%s_group.npz' % tag), flat=g_g, **v_g)
        np.savez(os.path.join(out_dir, '%s_local.npz' % tag), flat=g_l, **v_l)
        meta[tag] = m_g
    with open(os.path.join(out_dir, 'meta.json'), 'w') as f:
        json.dump(meta, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
