# coding: utf-8
"""Strip the optimizer state from a native checkpoint (the reference's misc/remove_optimizers_params_in_ckpt.py for its TF
checkpoints): train.py's `.npz` files hold, next to every variable, its Momentum / Adam / RMSProp slots, `optimizer/step` and
`global_step` (utils.misc_utils.Saver) - two to three times the size the forward needs.

    python misc/remove_optimizers_params_in_ckpt.py checkpoint/best_model.npz [--output shrinked_ckpt/shrinked.npz]
"""
import argparse
import os
import sys

import numpy as np

SLOT_SUFFIXES = ('/Momentum', '/Adam', '/Adam_1', '/RMSProp', '/RMSProp_1')
BOOKKEEPING = ('optimizer/step', 'global_step')


def shrink(src, dst):
    """Copies the variables of checkpoint `src` to `dst`; returns (kept keys, dropped keys)."""
    ckpt = np.load(src if src.endswith('.npz') else src + '.npz')
    drop = [k for k in ckpt.files if k in BOOKKEEPING or k.endswith(SLOT_SUFFIXES)]
    keep = [k for k in ckpt.files if k not in drop]
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    with open(dst if dst.endswith('.npz') else dst + '.npz', 'wb') as f:
        np.savez(f, **{k: ckpt[k] for k in keep})
    return keep, drop


def main(argv=None):
    ap = argparse.ArgumentParser(description="drop the optimizer slots from a native .npz checkpoint")
    ap.add_argument('ckpt_path')
    ap.add_argument('--output', default=os.path.join('shrinked_ckpt', 'shrinked.npz'))
    args = ap.parse_args(sys.argv[1:] if argv is None else argv)
    keep, drop = shrink(args.ckpt_path, args.output)
    print('%s: kept %d arrays, dropped %d optimizer arrays' % (args.output, len(keep), len(drop)))
    return keep, drop


if __name__ == '__main__':
    main()
