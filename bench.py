#!/usr/bin/env python
"""bench.py — the YOLOv3 hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Darknet-53 + 3-scale head forward (75 fused conv launches) on a
batch of 32 synthetic 416x416 fp32 images per GPU, random weights, input resident in HBM.  One "step" = one
forward over one batch.  Inference shards by image with no data-path collective: every rank
runs an independent replica on its own stream; torch.distributed (RCCL) is used only for the two barriers
and the max-over-ranks of the elapsed time.  Prints ONE JSON line on rank 0.

roofline: the dominant kernel is conv_mfma_f32_kernel<128,128,2,2,3,false,true> — the 3x3 implicit-GEMM MFMA
conv in its stream-K schedule (32 of the 75 conv layers, ~84 % of the FLOPs, ~70 % of the time).  It is
matrix-pipe bound in fp32: achieved = algorithmic FLOPs of those 32 launches / the sum of
their durations, measured with hipEvents recorded on the launch stream inside the timed region (one event per
layer boundary), peak = 157.3 TFLOP/s (fp32 MFMA).  In the default precision the dominant kernel is the Winograd
kernel conv_wino_f32_kernel<2,2> and `achieved` counts the algorithmic (direct) FLOPs.
fast_path: after the timed region the same steps are repeated with compute_dtype='f32_bf16x6' (fp32 tensors, every
product rebuilt from six bf16 matrix-pipe products, fp32 accumulation) and reported next to the
exact-fp32 `value` together with the max deviation between the two sets of feature maps.  It is never `value`.
cpu_baseline: the CPU oracle's torch-fp32 restatement of the same graph ("port"; the literal TF-CPU
reference cannot run here: no TensorFlow), same weights, a bounded sample, rank 0 and N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 32
SIZE = 416
CLASS_NUM = 80
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md chip-level table
ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326],
                   np.float32).reshape(9, 2)


def random_init(seed):
    """Random weights of the reference architecture with tame activations: He-normal kernels, BN close to
    identity, residual-branch gamma damped (so 23 residual adds do not blow the scale up)."""
    import torch
    import yolov3_tensorflow_amd as y3
    g = torch.Generator(device='cpu').manual_seed(seed)
    closers = set()
    idx = 2
    for blocks in (1, 2, 8, 8, 4):
        for _ in range(blocks):
            closers.add(idx + 1)
            idx += 2
        idx += 1
    for v in y3.global_variables(scope='yolov3'):
        parts = v.op_name.split('/')
        leaf = parts[-1]
        shape = tuple(v.shape)
        if leaf == 'weights':
            k, _, cin, cout = shape
            t = torch.randn(shape, generator=g) * float(np.sqrt(2.0 / (k * k * cin)))
            if cout == 3 * (5 + CLASS_NUM):
                t = t * 0.25
        elif leaf == 'gamma':
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
            conv = parts[-3]
            ci = 0 if conv == 'Conv' else int(conv.split('_')[1])
            if parts[-4] == 'darknet53_body' and ci in closers:
                t = t * 0.25
        elif leaf in ('beta', 'moving_mean'):
            t = torch.randn(shape, generator=g) * 0.05
        elif leaf == 'moving_variance':
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif leaf == 'biases':
            t = torch.randn(shape, generator=g) * 0.1
        else:
            raise AssertionError(v.op_name)
        v.assign(t)


def conv_flops(table, n, h, w):
    """Algorithmic FLOPs per layer for one forward (2*k^2*Cin*Cout*Hout*Wout*N), following the graph."""
    # spatial divisor of each layer's OUTPUT: replay the strides in creation order
    flops = []
    div = 1
    sdiv_of = []
    # body: divisors follow the stride-2 convs; head: 13-grid block, then 26, then 52
    head_divs = [32] * 8 + [16] * 8 + [8] * 7
    for i, (k, s, cin, cout, bn) in enumerate(table):
        if i < 52:
            div *= s
            d = div
        else:
            d = head_divs[i - 52]
        sdiv_of.append(d)
        flops.append(2.0 * k * k * cin * cout * (h // d) * (w // d) * n)
    return np.array(flops)


def traffic_from_profile(name='r01_pmc_traffic.json'):
    """HBM-side bytes per launch of the dominant kernel: rocprofv3 cannot run inside this process, so the
    number comes from the committed PMC passes (profiles/r01_pmc_traffic*.json: FETCH_SIZE x2 + WRITE_SIZE,
    collected over this same command); None if the file is absent."""
    path = os.path.join(ROOT, 'profiles', name)
    try:
        with open(path) as f:
            return int(json.load(f)['traffic_bytes_per_launch'])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(model_vars, budget_s=20.0):
    """Time the oracle's torch-CPU fp32 forward on a bounded sample (checker code, never the product)."""
    import torch
    from oracle import yolo_ref
    params = {v.op_name: v.numpy() for v in model_vars}
    x = np.random.RandomState(123).rand(2, SIZE, SIZE, 3).astype(np.float32)
    ncpu = os.cpu_count() or 1
    # oneDNN does not scale to every core of a big host on a 2-image batch: pick the best thread count
    best = None
    for nt in sorted({min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
        torch.set_num_threads(nt)
        yolo_ref.forward(params, x[:1])                   # warm-up (thread pool, oneDNN primitives)
        t0 = time.time()
        yolo_ref.forward(params, x)
        dt = time.time() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
    threads, per_batch = best
    torch.set_num_threads(threads)
    reps = int(max(1, min(10, budget_s / max(per_batch, 1e-3))))
    t0 = time.time()
    for _ in range(reps):
        yolo_ref.forward(params, x)
    dt = time.time() - t0
    return {"value": round(2 * reps / dt, 3), "unit": "images/s", "cores": int(threads), "kind": "port",
            "sample": "oracle.yolo_ref.forward (torch-CPU fp32 restatement of the reference graph; "
                      "TF-CPU itself is not installable here), %d x batch of 2 images %dx%d, same weights, "
                      "best of {64,32,16} threads on a %d-core host" % (reps, SIZE, SIZE, ncpu)}


PRECISION_TEXT = {
    "f32_wino": "exact fp32 MFMA arithmetic; Winograd F(2x2,3x3) kernel for the stride-1 3x3 convs, direct kernel elsewhere",
    "f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32), direct implicit-GEMM kernels only",
    "f32_bf16x6": "fp32 tensors; each product = 6 bf16 plane products, fp32 accumulate (dropped terms <= 2^-23 relative)",
    "f32_bf16x3": "fp32 tensors; each product = 3 bf16 plane products, fp32 accumulate (dropped terms <= 2^-15 relative)",
}


def measure_other_precision(model, dtype, primary, y3, x, fms, args, barrier, distributed, dist, world):
    """The bench workload once more with another compute_dtype (outside the timed region of `value`)."""
    import torch
    ref = [f.clone() for f in fms]
    model.compute_dtype = dtype
    try:
        with y3.variable_scope('yolov3'):
            for _ in range(args.warmup):
                fms2 = model.forward(x, False)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fms2 = model.forward(x, False)
            barrier()
            el2 = time.perf_counter() - t0
    finally:
        model.compute_dtype = primary
    if distributed:
        t = torch.tensor([el2], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el2 = float(t.item())
    return {"precision": dtype + ": " + PRECISION_TEXT[dtype],
            "value": round(world * BATCH * args.steps / el2, 2), "unit": "images/s",
            "ms_per_step": round(el2 / args.steps * 1e3, 4),
            "max_abs_diff_vs_primary": float(max((a - b).abs().max().item() for a, b in zip(ref, fms2))),
            "max_abs_feature": float(max(a.abs().max().item() for a in ref))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=['c2', 'c5'], default='c2',
                    help="c2 (default, the BASELINE metric): fp32 416x416 bs=32; c5: bf16-storage 608x608 bs=16")
    ap.add_argument('--precision', choices=['f32_wino', 'f32', 'f32_bf16x6', 'f32_bf16x3'], default='f32_wino',
                    help="c2 only. f32_wino (default): exact fp32 MFMA arithmetic, Winograd F(2x2,3x3) kernel for the "
                         "stride-1 3x3 convs, direct kernel elsewhere; f32: direct kernels only; f32_bf16x6 / "
                         "f32_bf16x3: fp32 tensors, every product rebuilt from 6 / 3 bf16 plane products with fp32 "
                         "accumulation")
    args = ap.parse_args()
    global BATCH, SIZE
    bf16 = args.workload == 'c5'
    if bf16:
        BATCH, SIZE = 16, 608

    import torch
    import torch.distributed as dist
    import yolov3_tensorflow_amd as y3
    from yolov3_tensorflow_amd import engine

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    y3.set_default_device('cuda:%d' % local_rank)
    distributed = world > 1 or 'RANK' in os.environ      # under torch.distributed.run even at N=1
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    model = y3.yolov3(CLASS_NUM, ANCHORS)
    if bf16:
        model.compute_dtype = 'bf16'
    split = (not bf16) and args.precision in ('f32_bf16x6', 'f32_bf16x3')
    wino = (not bf16) and args.precision == 'f32_wino'
    if not bf16:
        model.compute_dtype = args.precision
    x = torch.rand((BATCH, SIZE, SIZE, 3), device='cuda',
                   generator=torch.Generator(device='cuda').manual_seed(100 + rank))
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))      # create the variables
        random_init(seed=1)
        if wino:
            try:
                model.forward(x, False)
                torch.cuda.synchronize()
            except Exception as e:      # never lose the line to the newer kernel: fall back to the direct kernels
                print("bench.py: f32_wino failed (%s: %s); falling back to --precision f32" % (type(e).__name__, e),
                      file=sys.stderr)
                wino = False
                args.precision = 'f32'
                model.compute_dtype = 'f32'
        for _ in range(args.warmup):
            fms = model.forward(x, False)
        # per-layer hipEvents (for `roofline`) are recorded inside the timed region on every 4th step only: 150 event
        # records per step cost ~3 % of the step
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            model.set_layer_profiling(i % 4 == 0)
            fms = model.forward(x, False)
        barrier()
        elapsed = time.perf_counter() - t0
        layer_ms, table, main_ms, is_sk = model.read_layer_ms(with_main=True, shape=(BATCH, SIZE, SIZE))
        model.set_layer_profiling(False)
        # p50 of single-step latency (separate short loop; each step synchronised)
        lat = []
        for _ in range(min(args.steps, 10)):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.forward(x, False)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
    assert all(torch.isfinite(f).all().item() for f in fms), "non-finite feature maps"

    # Secondary measurement (c2, default precision only; outside the timed region above and never `value`): the same
    # workload with the products on the bf16 matrix pipe, and its deviation from the exact-fp32 feature maps.
    fast, direct = None, None
    if not bf16 and not split:
        primary = model.compute_dtype
        for key, dtype in (('fast', 'f32_bf16x6'), ('direct', 'f32')):
            if dtype == primary:
                continue
            try:
                res = measure_other_precision(model, dtype, primary, y3, x, fms, args, barrier, distributed, dist, world)
            except Exception as e:      # a secondary measurement must never cost the primary line
                res = {"error": "%s: %s" % (type(e).__name__, e)}
                model.compute_dtype = primary
            if key == 'fast':
                fast = res
            else:
                direct = res

    if distributed:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * BATCH * args.steps / elapsed
        flops = conv_flops(table, BATCH, SIZE, SIZE)
        is3 = np.array([k == 3 and cin != 3 for (k, s, cin, cout, bn) in table])
        sk_all = np.array(is_sk, bool)
        if wino:   # dominant family: the Winograd kernel (every stride-1 3x3 conv but the stem)
            is_sk = np.array([engine.wino_eligible(k, s, cin, cout)
                              for (k, s, cin, cout, bn) in table])
            main_ms = np.where(is_sk, layer_ms, main_ms)
        if bf16:   # dominant family: the 3x3 convs on 128x128 tiles (conv_mfma_bf16_kernel<128,128,2,2,3,false>)
            is_sk = np.array([k == 3 and cin != 3 and cout > 64 for (k, s, cin, cout, bn) in table])
        dom_ms = float(main_ms[is_sk].sum())          # the stream-K kernel alone (fix-up excluded)
        dom_flops = float(flops[is_sk].sum())
        n_dom = int(is_sk.sum())
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        # split precisions: fp32-equivalent peak = dense bf16 MFMA peak / products per fp32 multiply-add
        peak = 2500.0 if bf16 else {'f32': PEAK_FP32_MFMA_TFLOPS, 'f32_wino': PEAK_FP32_MFMA_TFLOPS,
                                    'f32_bf16x6': 2500.0 / 6,
                                    'f32_bf16x3': 2500.0 / 3}[args.precision]
        whole = float(flops.sum()) / (float(layer_ms.sum()) * 1e-3) / 1e12
        out = {
            "metric": ("images/sec at 608x608 bs=16 bf16 (Darknet-53 + 3-scale head forward)" if bf16 else
                       "images/sec at 416x416 bs=32 (Darknet-53 + 3-scale head forward)"),
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_image_p50": round(float(np.median(lat)) * 1e3 / BATCH, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32",
            "precision": ("bf16 storage, fp32 accumulate" if bf16 else PRECISION_TEXT[args.precision]),
            "data": "synthetic",
            "config": {"workload": ("configs[4]: Darknet-53 + 3-scale head forward, random weights, 608x608 bs=16 "
                                    "per GPU, bf16 storage / fp32 accumulate, input resident in HBM" if bf16 else
                                    "configs[1]: Darknet-53 + 3-scale head forward, random weights, "
                                    "416x416 bs=32 fp32 per GPU, input resident in HBM"),
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "image_size": SIZE,
                       "class_num": CLASS_NUM, "parallelism": "replicas (image-sharded, no collective)"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": (None if (bf16 or split) else
                                     traffic_from_profile('r01_pmc_traffic_wino.json' if wino else 'r01_pmc_traffic.json')),
                         "kernel": ("conv_mfma_bf16_kernel<128,128,2,2,3,false> (3x3 implicit-GEMM conv, bf16 storage; "
                                    "staging-bound, see DESIGN.md)" if bf16 else
                                    "conv_mfma_split_kernel<128,128,2,2,3,false,true,%d,false> (3x3 implicit-GEMM "
                                    "conv on the bf16 matrix pipe, stream-K schedule; peak = 2500/%d fp32-equivalent)"
                                    % ((3, 6) if args.precision == 'f32_bf16x6' else (2, 3)) if split else
                                    "conv_wino_f32_kernel<2,2> (Winograd F(2x2,3x3) conv, fp32 MFMA; `achieved` counts the "
                                    "ALGORITHMIC (direct-convolution) FLOPs, so frac can exceed 1: the kernel issues "
                                    "1/2.25 of them as MFMA work, see mfma_work_frac)" if wino else
                                    "conv_mfma_f32_kernel<128,128,2,2,3,false,true,false> (3x3 implicit-GEMM conv, "
                                    "stream-K schedule)"),
                         "launches_per_step": n_dom,
                         "avg_launch_ms": round(dom_ms / n_dom, 4),
                         "algorithmic_gflop_per_launch": round(dom_flops / n_dom / 1e9, 3),
                         "fixup_ms_per_step": round(float((layer_ms - main_ms)[sk_all].sum()), 4),
                         "all_3x3_tflops": round(float(flops[is3].sum()) / (float(layer_ms[is3].sum()) * 1e-3) / 1e12, 2),
                         "whole_forward_tflops": round(whole, 2),
                         "sum_layer_ms": round(float(layer_ms.sum()), 4)},
        }
        if bf16:
            # algorithmic HBM traffic of the whole bf16 forward: 6.565 GB per bs=16 batch at 608
            out["roofline"]["whole_forward_hbm_tbps"] = round(6.565e9 / (ms_per_step * 1e-3) / 1e12, 3)
        if wino:
            out["roofline"]["mfma_work_tflops"] = round(achieved / 2.25, 2)
            out["roofline"]["mfma_work_frac"] = round(achieved / 2.25 / peak, 4)
        if direct is not None:
            out["direct_path"] = direct
        if fast is not None:
            out["fast_path"] = fast
        if world == 1 and not args.no_cpu_baseline and not bf16:
            out["cpu_baseline"] = cpu_baseline(y3.global_variables(scope='yolov3'))
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
