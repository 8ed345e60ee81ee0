#!/usr/bin/env python
"""bench.py — the YOLOv3 hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|c5|feeder]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself under torch.distributed.run
(one rank per GPU, rendezvous on 127.0.0.1).

Workloads
  c2 (default; BASELINE.json configs[1], the metric): Darknet-53 + 3-scale head forward (75 fused conv launches) on a
     batch of 32 synthetic 416x416 fp32 images per GPU, random weights, input resident in HBM.  One "step" = one forward
     over one batch.  Inference shards by image with no data-path collective: every rank runs an
     independent replica on its own stream; torch.distributed (RCCL) carries only the two barriers and the max-over-ranks
     of the elapsed time.
  c4 (configs[3]): one TRAIN step — forward(is_training) -> compute_loss -> backward -> bucketed RCCL all-reduce of the
     gradients (overlapped with backward) -> clip -> SGD update of the whole model — at 416x416, bs=64 per GPU.  `value` is
     measured on resident synthetic tensors; `fed` repeats the steps with the batches coming from the feeder (JPEG decode,
     the reference's augmentation chain, resize, upload, target assignment under the steps).
  c5 (configs[4]): the c2 forward with bf16 storage at 608x608, bs=16 per GPU.
  feeder: the host side of c4 alone - one feeder per rank (10 worker threads, prefetch 5: the reference's defaults) decoding
     synthetic JPEGs, augmenting, resizing to 416x416 and uploading bs=64 batches that nothing consumes (run_feeder).
Prints ONE JSON line on rank 0.
Inference workloads (c2, c5, detect) run the batch as --streams equal parts on that many HIP streams of the GPU (default 2:
model.inference_streams; BASELINE north star "independent per-GPU streams for inference") - except the forwards that carry
per-layer hipEvents (every 8th step of the timed region), which run on one stream: `roofline` is computed from those, i.e.
from undisturbed whole-batch launches.  `--streams 1` is the one-stream command (what the rocprof kernel statistics under
profiles/ are taken with, so that their per-launch averages are comparable with `roofline.avg_launch_ms`).

roofline (c2): the forward is bound by the fp32 matrix pipe.  The dominant kernel family is the two
Winograd kernels of the stride-1 3x3 convs (32 launches, ~60 % of the step): conv_wino44_f32_kernel (F(4x4,3x3): the
convs with Cin >= 64 when the launch fills the chip) and conv_wino8_f32_kernel (F(2x2,3x3): the others).
`achieved` = the useful MFMA work of those launches - per layer 16/36 (F(2x2)) or 36/144 (F(4x4); the zero tiles that pad
a 13- / 26-grid to a multiple of 4 are issued but NOT counted) of the direct-convolution FLOPs, by the kernel the library
picks (winograd_issue_factors) - / the sum of their durations,
measured with hipEvents recorded on the launch stream inside the timed region; `peak` = 157.3 TFLOP/s (fp32 MFMA);
`frac` = achieved / peak.  `achieved_algorithmic` counts the direct-convolution FLOPs (it may exceed `peak`: Winograd
needs 2.25x / 4x fewer multiplies).  `whole_forward_frac` = sum over the 75 layers of max(bytes / 8 TB/s, issued FLOPs /
157.3 TF/s), divided by the measured ms_per_step.  `traffic`: fabric-side bytes per launch from a committed rocprofv3
--pmc pass, reported only when that pass was taken on the kernel sources of this build.
fast_path / direct_path: after the timed region the same steps are repeated with other compute_dtypes and reported next
to `value` with the max deviation between the feature maps.  Never `value`.
cpu_baseline: the CPU oracle's torch-fp32 restatement of the same graph ("port"; the literal TF-CPU reference cannot run
here: no TensorFlow), same weights, ONE bounded sample, rank 0 and N=1 only.
box_delta_vs_oracle: outside the timed region, the decoded boxes/confs/probs of all 32 images of the bench batch against
the CPU oracle in fp32 and fp64 (BASELINE metric: "box delta vs ref"), with the fp32 oracle's own drift beside them.
detect / c5 / c4 / feeder: the other BASELINE configurations, and the host side of c4, as secondary objects of the c2 line
(measured after everything above).
"""
import os as _os
# Two HIP streams per GPU only overlap when they sit on different hardware queues; with the runtime's default of 4 queues and
# RCCL initialised in the process the side stream of the inference workloads was seen to alias (12.1 ms against 10.4): ask for
# 8 before HIP starts (a no-op if the caller already set it), and let choose_inference_streams() measure anyway.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import argparse
import json
import os
import socket
import subprocess
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
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_TBPS = 8.0
ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326],
                   np.float32).reshape(9, 2)


SECONDARY_BUDGET_S = 420      # wall-clock allowance for the detect / c5 / c4 secondaries of the default line (measured: ~70 s at N = 1)


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


def layer_divisors(table):
    """Spatial divisor of each layer's OUTPUT relative to the network input."""
    head_divs = [32] * 8 + [16] * 8 + [8] * 7
    out, div = [], 1
    for i, (k, s, cin, cout, bn) in enumerate(table):
        if i < 52:
            div *= s
            out.append(div)
        else:
            out.append(head_divs[i - 52])
    return out


def conv_flops(table, n, h, w):
    """Algorithmic (direct-convolution) FLOPs per layer for one forward: 2*k^2*Cin*Cout*Hout*Wout*N."""
    return np.array([2.0 * k * k * cin * cout * (h // d) * (w // d) * n
                     for (k, s, cin, cout, bn), d in zip(table, layer_divisors(table))])


def conv_bytes(table, n, h, w, esize=4, fused=None):
    """Algorithmic HBM bytes per layer: input + packed kernel + output (+ the residual the 3x3 of a
    res_block adds).  The upsampled/concatenated input of the two head convs is counted at its stored size.
    `fused` (model.layer_fused): a layer that runs inside the next layer's launch (1) neither writes its output nor has the
    next layer read it - the tensor between them never reaches memory -, and a fused residual block reads its input once (it is
    the 1x1 conv's operand and the shortcut): the bytes of a fused pair are those of ONE kernel, booked on the launching layer."""
    out = []
    resid = set()
    idx = 2
    for blocks in (1, 2, 8, 8, 4):
        for _ in range(blocks):
            resid.add(idx + 1)
            idx += 2
        idx += 1
    for i, ((k, s, cin, cout, bn), d) in enumerate(zip(table, layer_divisors(table))):
        ho, wo = h // d, w // d
        hi, wi = ho * s, wo * s
        b_in, b_w, b_out = n * hi * wi * cin * esize, k * k * cin * cout * 4, n * ho * wo * cout * esize
        if i == 0:
            b_in = n * hi * wi * cin * 4               # the image is fp32 in every mode
        b_res = b_out if i in resid else 0
        if fused is not None and fused[i] == 1:        # runs inside the next launch: its output never reaches memory
            b_out = 0
        if fused is not None and fused[i] == 2:        # ... whose input is that tensor; a fused block's shortcut is its first input
            b_in, b_res = 0, 0
        out.append(float(b_in + b_w + b_out + b_res))
    return np.array(out)


def traffic_from_profile(names):
    """HBM-side bytes per launch of the dominant kernel from a committed PMC pass (profiles/*.json: the byte-weighted
    TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B counters over one forward of this workload,
    tools/pmc_layers.py + tools/pmc_traffic_layers.py).  rocprofv3 cannot run inside this process, so the figure is
    static - it is reported only when the file carries the hash of THIS build's kernel sources (`csrc_sha16`); a file
    taken on other sources gives traffic = null and says so.  Returns (bytes or None, source)."""
    from yolov3_tensorflow_amd.build import csrc_sha16
    here = csrc_sha16()
    stale = None
    for name in names:
        path = os.path.join(ROOT, 'profiles', name)
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get('csrc_sha16') == here:
                return int(d['traffic_bytes_per_launch']), ('profiles/%s (separate rocprofv3 --pmc pass on this build, '
                                                            'csrc %s)' % (name, here))
            stale = stale or ('profiles/%s was taken on kernel sources %s, this build is %s: not reported'
                              % (name, d.get('csrc_sha16', '(unstamped)'), here))
        except (OSError, KeyError, ValueError):
            continue
    return None, stale


def library_stamp():
    """Which HIP library this process measured (Y3_LIB_PATH lets tools load a probe build): the path ctypes loaded,
    whether an override was in force, and the hash of the kernel sources in the tree beside this file."""
    from yolov3_tensorflow_amd import _lib
    from yolov3_tensorflow_amd.build import csrc_sha16
    return {"lib_path": os.path.relpath(_lib.LIB_PATH, ROOT) if _lib.LIB_PATH.startswith(ROOT) else _lib.LIB_PATH,
            "override_env_Y3_LIB_PATH": bool(os.environ.get("Y3_LIB_PATH")), "csrc_sha16": csrc_sha16()}


def cpu_quota_cores():
    """Cores this process may actually use: the cgroup CPU quota when there is one (the GPU boxes show 256 logical CPUs and
    cpu.max = 16 cores), else the logical CPU count."""
    ncpu = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            return max(1, min(ncpu, int(round(int(quota) / float(period))))), ncpu
    except (OSError, ValueError):
        pass
    return ncpu, ncpu


def cpu_baseline(model_vars, budget_s=12.0):
    """Time the oracle's torch-CPU fp32 forward (checker code, never the product) on ONE bounded sample: batches of 2
    images on min(16, CPU quota) threads for about `budget_s` seconds.  The GPU boxes show 256 logical CPUs but their cgroup
    grants 16 cores (cpu.max = 1600000 100000, found in round 4): that is why 16 threads measured best in round 2 (16-21
    images/s) and 64 / 256 threads only oversubscribed (3.2 / 0.26 images/s)."""
    import torch
    from oracle import yolo_ref
    params = {v.op_name: v.numpy() for v in model_vars}
    quota, ncpu = cpu_quota_cores()
    threads = min(quota, 16)
    x = np.random.RandomState(123).rand(2, SIZE, SIZE, 3).astype(np.float32)
    torch.set_num_threads(threads)
    yolo_ref.forward(params, x[:1])                     # warm-up (thread pool, oneDNN primitives)
    t0 = time.time()
    yolo_ref.forward(params, x)
    per_batch = time.time() - t0
    reps = int(max(1, min(60, budget_s / max(per_batch, 1e-3))))
    t0 = time.time()
    for _ in range(reps):
        yolo_ref.forward(params, x)
    dt = time.time() - t0
    return {"value": round(2 * reps / dt, 3), "unit": "images/s", "cores": int(threads), "cores_note": "%d threads; the process may use %d cores (cgroup cpu.max) of the host's %d logical "
            "CPUs - more threads than the quota only oversubscribe (bs=32 on 64 threads measured 3.2 images/s)"
            % (threads, quota, ncpu), "kind": "port",
            "sample": "oracle.yolo_ref.forward (torch-CPU fp32 restatement of the reference graph; TF-CPU itself is "
                      "not installable here), %dx%d, same weights; %d batches of 2 images on %d threads of the %d-core "
                      "host (%.1f s)" % (SIZE, SIZE, reps, threads, ncpu, dt)}


def box_delta_vs_oracle(model, y3, x, fms, n_check=None, chunk=4):
    """Decoded boxes / confs / probs of EVERY image of the bench batch (taken from the batch's own feature maps) against
    the CPU oracle's forward + predict on those images (checker; outside the timed region), and beside it the drift of
    the fp32 CPU oracle itself against the fp64 CPU oracle on the same images: the north star's 1e-3 is met relative to
    the box scale, and the absolute pixel figure (w = exp(t_w) * anchor amplifies an fp32 logit drift) is the fp32
    floor, not the kernels - the two columns make that visible."""
    import torch
    from oracle import yolo_ref
    params = {v.op_name: v.numpy() for v in y3.global_variables(scope='yolov3')}
    n_all = int(x.shape[0])
    n_check = n_all if n_check is None else min(int(n_check), n_all)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    boxes, confs, probs = model.predict([f[:n_check].contiguous() for f in fms])
    gb, gc, gp = boxes.cpu().numpy(), confs.cpu().numpy(), probs.cpu().numpy()
    gf = [f[:n_check].cpu().numpy() for f in fms]

    def delta(ab, ac, ap, rb, rc, rp):
        scale = np.maximum(np.abs(rb).max(axis=-1, keepdims=True), 1.0)
        inside = (np.abs(rb).max(axis=-1) <= 2.0 * SIZE)
        d = np.abs(ab - rb)
        return [float((d / scale).max()), float(d[inside].max()) if inside.any() else 0.0,
                float(np.abs(ac - rc).max()), float(np.abs(ap - rp).max())]

    gpu_vs_32 = np.zeros(4)
    gpu_vs_64 = np.zeros(4)
    o32_vs_64 = np.zeros(4)
    fm_err32 = fm_err64 = fm_o32_64 = 0.0
    for i in range(0, n_check, chunk):
        xs = x[i:i + chunk].cpu().numpy()
        r32 = yolo_ref.forward(params, xs)
        r64 = yolo_ref.forward(params, xs, dtype=torch.float64)
        p32 = yolo_ref.predict(r32, ANCHORS, [SIZE, SIZE], CLASS_NUM)
        p64 = yolo_ref.predict([np.asarray(r, np.float64) for r in r64], ANCHORS, [SIZE, SIZE], CLASS_NUM,
                               dtype=np.float64)
        sl = slice(i, i + chunk)
        gpu_vs_32 = np.maximum(gpu_vs_32, delta(gb[sl], gc[sl], gp[sl], *p32))
        gpu_vs_64 = np.maximum(gpu_vs_64, delta(gb[sl], gc[sl], gp[sl], *p64))
        o32_vs_64 = np.maximum(o32_vs_64, delta(p32[0], p32[1], p32[2], *p64))
        fm_err32 = max(fm_err32, max(float(np.abs(g[sl] - r).max()) for g, r in zip(gf, r32)))
        fm_err64 = max(fm_err64, max(float(np.abs(g[sl] - r).max()) for g, r in zip(gf, r64)))
        fm_o32_64 = max(fm_o32_64, max(float(np.abs(a - r).max()) for a, r in zip(r32, r64)))
    keys = ("boxes_max_rel_to_box_scale", "boxes_max_abs_px_within_2x_image", "confs_max_abs", "probs_max_abs")
    out = {"images": int(n_check)}
    out.update({k: float(v) for k, v in zip(keys, gpu_vs_32)})
    out["feature_maps_max_abs"] = fm_err32
    out["tolerance"] = "1e-3 (north star), read relative to the box scale"
    out["oracle"] = "oracle.yolo_ref (torch-CPU fp32)"
    out["gpu_vs_fp64_oracle"] = dict({k: float(v) for k, v in zip(keys, gpu_vs_64)}, feature_maps_max_abs=fm_err64)
    out["fp32_oracle_vs_fp64_oracle"] = dict({k: float(v) for k, v in zip(keys, o32_vs_64)},
                                             feature_maps_max_abs=fm_o32_64,
                                             note="drift of the CPU fp32 restatement itself on the same images: the "
                                                  "fp32 floor the GPU figures should be read against")
    return out


PRECISION_TEXT = {
    "f32_wino": "fp32 MFMA arithmetic throughout; Winograd kernels for the stride-1 3x3 convs (F(4x4,3x3) where the library's "
                "y3_conv_wino44_preferred says so: the 31 convs with Cin >= 64 at this size; F(2x2,3x3) for the 32->64 one), direct "
                "kernels elsewhere",
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


def synthetic_y_true(n, size, class_num, anchors, seed, device):
    """Device-side stand-in for process_box output (utils/data_utils.py:51-115 layout): three object cells per image
    and scale, class uniform, box = the cell centre with the anchor's size, mix weight 1."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    out = []
    for s, a0 in ((32, 6), (16, 3), (8, 0)):
        gsz = size // s
        y = torch.zeros((n, gsz, gsz, 3, 6 + class_num))
        y[..., -1] = 1.0
        cy = torch.randint(0, gsz, (n, 3), generator=g)
        cx = torch.randint(0, gsz, (n, 3), generator=g)
        ka = torch.randint(0, 3, (n, 3), generator=g)
        cl = torch.randint(0, class_num, (n, 3), generator=g)
        for i in range(n):
            for j in range(3):
                k = int(ka[i, j])
                yy, xx = int(cy[i, j]), int(cx[i, j])
                y[i, yy, xx, k, 0:4] = torch.tensor([(xx + 0.5) * s, (yy + 0.5) * s, float(anchors[a0 + k][0]),
                                                     float(anchors[a0 + k][1])])
                y[i, yy, xx, k, 4] = 1.0
                y[i, yy, xx, k, 5 + int(cl[i, j])] = 1.0
        out.append(y.to(device))
    return out


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: run N ranks of this script under torch.distributed.run."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % args.gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    print('bench.py: launching %d ranks: %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL (see the environment notes)
    return subprocess.call(cmd, env=env)


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--streams', type=int, default=2,
                    help="HIP streams per GPU for the inference workloads (c2, c5, detect): the batch runs as that many equal "
                         "parts, one per stream (model.inference_streams; forwards with per-layer events on stay on one stream)")
    ap.add_argument('--no-secondary', action='store_true',
                    help="c2 only: skip the secondary objects (fast_path / direct_path, detect, c5, c4, feeder) measured after the "
                         "timed region of `value`")
    ap.add_argument('--workload', choices=['c2', 'c4', 'c5', 'feeder'], default='c2',
                    help="c2 (default, the BASELINE metric): fp32 forward 416x416 bs=32; c4: train step 416x416 bs=64 "
                         "per GPU, SGD, RCCL gradient all-reduce; c5: bf16-storage forward 608x608 bs=16; feeder: the "
                         "host side of c4 alone (decode + augment + resize + upload of bs=64 batches, nothing consuming)")
    ap.add_argument('--precision', choices=['f32_wino', 'f32', 'f32_bf16x6', 'f32_bf16x3'], default='f32_wino',
                    help="c2/c4. f32_wino (default): fp32 MFMA arithmetic, Winograd kernels for the stride-1 3x3 convs "
                         "(forward: F(4x4,3x3) where the library prefers it, F(2x2,3x3) elsewhere; train step: F(2x2,3x3)), "
                         "direct kernel elsewhere; f32: direct kernels only; f32_bf16x6 / "
                         "f32_bf16x3: fp32 tensors, every product rebuilt from 6 / 3 bf16 plane products with fp32 "
                         "accumulation")
    ap.add_argument('--batch', type=int, default=None, help="per-GPU batch (default: the workload's BASELINE value)")
    ap.add_argument('--wgrad-stream', type=int, choices=[0, 1], default=None,
                    help="c4: backward's weight gradients on a second stream (1) or on the main one (0); default: the package's")
    ap.add_argument('--no-fed', action='store_true', help="c4: skip the repetition of the steps on batches from the feeder")
    ap.add_argument('--head-only', action='store_true', help="c4: update only yolov3/yolov3_head (the reference's "
                                                             "default update_part) instead of the whole model")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 5 if args.workload == 'c4' else 20
    if args.warmup is None:
        args.warmup = 2 if args.workload == 'c4' else 5
    return args


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args, argv))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('Y3_BENCH_DRY_RUN') == '1':      # CPU-side check of the launch plumbing (tests)
        print('bench.py: dry-run rank %d/%d (local %d), workload %s' % (rank, world, local_rank, args.workload), flush=True)
        return
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running with WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    def set_workload(name, batch=None):
        global BATCH, SIZE
        BATCH, SIZE = {'c2': (32, 416), 'c4': (64, 416), 'c5': (16, 608), 'feeder': (64, 416)}[name]
        if batch:
            BATCH = batch

    set_workload(args.workload, args.batch)

    import torch
    import torch.distributed as dist
    import yolov3_tensorflow_amd as y3

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

    per_rank = {}       # name -> every rank's own seconds for the last timed region (reported as per_rank_ms_per_step)

    def max_over_ranks(seconds):
        if not distributed:
            per_rank['last'] = [seconds]
            return seconds
        t = torch.tensor([seconds], device='cuda', dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank['last'] = [float(v.item()) for v in every]
        return max(per_rank['last'])
    max_over_ranks.per_rank = per_rank

    if args.workload == 'c4':
        out = run_train(args, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks)
    elif args.workload == 'feeder':
        out = run_feeder(args, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks)
    else:
        out = run_forward(args, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks)
    if args.workload == 'c2' and not args.no_secondary and args.batch is None:
        # The other BASELINE configurations, measured by the same (driver-run) command AFTER the timed region of `value`
        # and never part of it: detect = forward + decode + gpu_nms (what test_single_image.py / eval.py run), c5 =
        # configs[4] (608x608 bf16 bs=16), c4 = configs[3] (train step, bs=64 per GPU, RCCL all-reduce when N > 1).
        # Every rank runs them (c4 has a collective); a failure costs only that object.
        import copy
        import threading
        # Watchdog: the secondaries run collectives at N > 1 (barriers, the c4 gradient all-reduce); should one of them
        # ever stall (a rank that failed alone leaves the others waiting), the primary line must still come out: after
        # SECONDARY_BUDGET_S rank 0 prints what it has and every rank leaves with status 0.
        def give_up():
            if rank == 0 and out is not None:
                for name in ('detect', 'c5', 'c4', 'feeder'):
                    out.setdefault(name, {"error": "not finished within %d s (watchdog)" % SECONDARY_BUDGET_S})
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(SECONDARY_BUDGET_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        for name, fn in (('detect', run_detect), ('c5', run_forward), ('c4', run_train), ('feeder', run_feeder)):
            sub = copy.copy(args)
            sub.workload = name if name != 'detect' else 'c2'
            sub.no_secondary, sub.no_cpu_baseline, sub.secondary = True, True, True
            sub.steps, sub.warmup = (8, 3) if name in ('c4', 'feeder') else (10, 3)
            set_workload(sub.workload)
            try:
                torch.cuda.empty_cache()
                res = fn(sub, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks)
            except Exception as e:      # a secondary measurement must never cost the primary line
                res = {"error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0 and out is not None:
                out[name] = slim(res)
        watchdog.cancel()
        set_workload(args.workload, args.batch)
    if rank == 0:
        if isinstance(out, dict):
            out["library"] = library_stamp()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def run_feeder(args, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks):
    """SURVEY.md 8f row 1, the host side of the train step on its own: every rank's feeder (reference train.py:34-43:
    num_parallel_calls = 10, prefetch 5) decodes synthetic 640x480 JPEGs, runs the reference's 'train' augmentation chain
    with mix-up, resizes to 416x416, fills pinned buffers and uploads bs=64 batches - with nothing consuming them but this
    loop, so the figure is what the host side can deliver, to be read against what the c4 step consumes."""
    import importlib.util
    import pathlib
    import tempfile
    spec = importlib.util.spec_from_file_location('feeder_rate', os.path.join(ROOT, 'tools', 'feeder_rate.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    from yolov3_tensorflow_amd.feeder import Feeder
    workers, prefetch = 10, 5
    total = (args.steps + args.warmup + 1) * BATCH
    with tempfile.TemporaryDirectory() as folder:
        lines = tool.write_set(pathlib.Path(folder), 256, seed=rank)
        lines = (lines * (total // len(lines) + 1))[:total]
        feeder = Feeder(lines, BATCH, CLASS_NUM, [SIZE, SIZE], tool.ANCHORS, mode='train', use_mix_up=True,
                        num_threads=workers, prefetch=prefetch, seed=1 + rank)
        it = feeder.epoch(0)
        for _ in range(args.warmup):
            next(it)
        barrier()
        t0, c0 = time.perf_counter(), time.process_time()
        for _ in range(args.steps):
            next(it)
        barrier()
        elapsed, cpu = max_over_ranks(time.perf_counter() - t0), time.process_time() - c0
        it.close()
        feeder.close()
    if rank != 0:
        return None
    return {
        "metric": "images/sec delivered by the feeder (decode + 'train' augmentation with mix-up + resize to 416x416 + "
                  "upload), nothing consuming",
        "value": round(world * BATCH * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic 640x480 JPEGs, 1-7 boxes each",
        "config": {"workload": "the host side of configs[3]: per GPU one feeder, %d worker threads, prefetch %d, bs=%d, "
                               "pixel work %s" % (workers, prefetch, BATCH,
                                                  "on the device (y3_feed_run; planned by liby3feed.so)"
                                                  if feeder.pixels == 'gpu' else "in liby3feed.so"),
                   "batch_per_gpu": BATCH, "image_size": SIZE, "backend": feeder.backend, "workers": workers,
                   "pixels": feeder.pixels},
        "host_cpu_ms_per_image": round(1e3 * cpu / (BATCH * args.steps), 3),      # rank 0's process, all threads
    }


def slim(res):
    """A secondary object of the c2 line: the measurement, its workload and its roofline (no nested secondaries)."""
    if res is None or 'error' in res:
        return res
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype', 'precision', 'scaling',
            'config', 'roofline', 'loss', 'peak_mem_gb', 'regimes', 'data', 'fed', 'host_cpu_ms_per_image', 'wgrad_stream')
    return {k: res[k] for k in keep if k in res}


# ------------------------------------------------------------------------------------------------------------
# detect: forward + decode + per-class NMS (what test_single_image.py:48-62 and eval.py:96-123 run per image)
# ------------------------------------------------------------------------------------------------------------
def run_detect(args, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks):
    """yolov3.detect() on the c2 batch (416x416, bs=32): y3_net_forward -> y3_decode (+ fused conf*prob) -> y3_nms for all
    images, everything resident on the device.  Two reference parameter sets: test_single_image.py:55 (max_boxes 200,
    score 0.3, IoU 0.45) and eval.py:47-54 (400, 0.01, 0.45); two score regimes: 'detector' = the random head with the
    objectness biases shifted by -4.6 and the class biases by -3 (a 1 % objectness and a 5 % class prior, so the candidates
    above either score threshold are sparse, as for a trained detector) and
    'dense' = the unshifted random head (conf ~ prob ~ 0.5: nearly every one of the 10,647 x 80 scores passes 0.01 - the
    worst case of the greedy per-class NMS)."""
    from yolov3_tensorflow_amd import framework as fw
    model = y3.yolov3(CLASS_NUM, ANCHORS)
    model.inference_streams = max(1, int(getattr(args, 'streams', 1)))
    model.compute_dtype = args.precision
    x = torch.rand((BATCH, SIZE, SIZE, 3), device='cuda',
                   generator=torch.Generator(device='cuda').manual_seed(100 + rank))
    params = (('test_single_image', 200, 0.3, 0.45), ('eval', 400, 0.01, 0.45))
    regimes = {}
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
        random_init(seed=1)
        if model.inference_streams > 1:
            model.choose_inference_streams(x, candidates=(1, model.inference_streams))
        heads = [v for v in y3.global_variables(scope='yolov3/yolov3_head') if v.op_name.endswith('/biases')]
        for regime in ('detector', 'dense'):
            saved = [v.tensor.clone() for v in heads]
            if regime == 'detector':
                for v in heads:
                    t = v.tensor.clone().view(3, 5 + CLASS_NUM)
                    t[:, 4] -= 4.6
                    t[:, 5:] -= 3.0
                    v.assign(t.view(-1))
            try:
                for name, max_boxes, score_t, iou_t in params:
                    for _ in range(args.warmup):
                        dets = model.detect(x, max_boxes, score_t, iou_t)
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        dets = model.detect(x, max_boxes, score_t, iou_t)
                    barrier()
                    el = max_over_ranks(time.perf_counter() - t0)
                    fw.check_context()
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        model.forward(x, False)
                    barrier()
                    el_fwd = max_over_ranks(time.perf_counter() - t0)
                    regimes['%s/%s' % (regime, name)] = {
                        "max_boxes": max_boxes, "score_thresh": score_t, "nms_thresh": iou_t,
                        "images_per_s": round(world * BATCH * args.steps / el, 2),
                        "ms_per_batch": round(el / args.steps * 1e3, 4),
                        "forward_ms": round(el_fwd / args.steps * 1e3, 4),
                        "decode_plus_nms_ms": round((el - el_fwd) / args.steps * 1e3, 4),
                        "detections_per_image": round(float(np.mean([int(d[0].shape[0]) for d in dets])), 1)}
            finally:
                for v, t in zip(heads, saved):
                    v.assign(t)
    if rank != 0:
        return None
    head = regimes['detector/test_single_image']
    # bound: the forward's per-layer bound + one pass over the feature maps and the decoded tensors at the HBM rate
    table = [tuple(t) for t in model._get_net(x.device)['table']]
    flops = conv_flops(table, BATCH, SIZE, SIZE)
    nbytes = conv_bytes(table, BATCH, SIZE, SIZE, 4)
    is_wino, fac = winograd_issue_factors(table, args.precision, SIZE)
    issued = flops * fac
    fwd_bound = float(np.maximum(nbytes / (PEAK_HBM_TBPS * 1e12), issued / (PEAK_FP32_MFMA_TFLOPS * 1e12)).sum()) * 1e3
    nbox = 3 * sum((SIZE // s) ** 2 for s in (32, 16, 8))
    post_bytes = BATCH * nbox * (5 + CLASS_NUM) * 4 * 2 + BATCH * nbox * CLASS_NUM * 4 * 2   # decode in/out, scores w + r
    post_bound = post_bytes / (PEAK_HBM_TBPS * 1e12) * 1e3
    return {"metric": "images/sec, forward + decode + per-class gpu_nms at 416x416 bs=%d (yolov3.detect)" % BATCH,
            "value": head["images_per_s"], "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_batch"], "dtype": "f32",
            "precision": PRECISION_TEXT[args.precision], "scaling": "weak",
            "config": {"workload": "configs[2]-shaped: the c2 forward + y3_decode + y3_nms (TF semantics), random "
                                   "weights, 416x416 bs=%d; `value` = detector-like scores with test_single_image.py's "
                                   "parameters (200 / 0.3 / 0.45); all four (regime, parameter set) pairs in `regimes`"
                                   % BATCH, "streams_per_gpu": model.inference_streams},
            "roofline": {"bound": "mfma+hbm", "unit": "ms",
                         "peak": round(fwd_bound + post_bound, 4), "achieved": head["ms_per_batch"],
                         "frac": round((fwd_bound + post_bound) / head["ms_per_batch"], 4),
                         "traffic": None,
                         "kernel": "whole pipeline: sum over the 75 conv layers of max(bytes / 8 TB/s, issued FLOPs / "
                                   "157.3 TF/s) + one pass over the feature maps, decoded tensors and scores at 8 TB/s "
                                   "(%.3f + %.3f ms) over the measured ms per batch" % (fwd_bound, post_bound)},
            "regimes": regimes}


def layer_input_grids(table, size):
    """Input grid (pixels per side) of each of the 75 convs: the body halves it at every stride-2 conv (416 -> 13), the
    head runs at 13, 26 (from layer 60 on) and 52 (from layer 68 on)."""
    grids, g = [], size
    for li, (k, s, cin, cout, bn) in enumerate(table):
        if li == 60 or li == 68:
            g *= 2
        grids.append(g)
        if s == 2:
            g //= 2
    return grids


def winograd_issue_factors(table, precision, size):
    """Per layer: (is_winograd, MFMA work ISSUED / direct-convolution FLOPs) for the kernel the library runs in `precision`.
    F(2x2,3x3) layers issue 16/36 of the direct count; the layers y3_conv_wino44_preferred names run F(4x4,3x3) in the
    inference forward: 36/144.  The zero pixels of the 4x4 tiles that pad a 13- or a 26-grid (image by image x1.51 / x1.16; with
    the batch tiled as one mosaic, csrc/y3_conv_wino44.hip w44_tiling, x1.16 / x1.08 at bs=32) ARE issued by the kernel and are
    NOT counted (multiplying zeros is not achieved work), so `frac` is the useful fraction.
    The factor follows the library's own choice of kernel per layer."""
    from yolov3_tensorflow_amd import engine
    is_w, fac = [], []
    grid_of = layer_input_grids(table, size)
    for li, (k, s, cin, cout, bn) in enumerate(table):
        g_in = grid_of[li]
        w = precision == 'f32_wino' and bool(engine.wino_eligible(k, s, cin, cout))
        f = 1.0
        if w:
            if engine.wino44_preferred(BATCH, g_in, g_in, k, s, cin, cout):
                f = 36.0 / 144.0          # USEFUL work only: the zero tiles a 13- or 26-grid is padded with are not counted
            else:
                f = 16.0 / 36.0
        is_w.append(w)
        fac.append(f)
    return np.array(is_w, bool), np.array(fac)


def engine_wino(k, s, cin, cout):
    from yolov3_tensorflow_amd import engine
    return bool(engine.wino_eligible(k, s, cin, cout))


# ------------------------------------------------------------------------------------------------------------
# c4: the train step
# ------------------------------------------------------------------------------------------------------------
def run_train(args, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks):
    from yolov3_tensorflow_amd import engine, training, framework as fw
    from yolov3_tensorflow_amd.utils.misc_utils import config_optimizer
    model = y3.yolov3(CLASS_NUM, ANCHORS, batch_norm_decay=0.99, weight_decay=5e-4)
    model.compute_dtype = args.precision
    x = torch.rand((BATCH, SIZE, SIZE, 3), device='cuda', generator=torch.Generator(device='cuda').manual_seed(100 + rank))
    yt = synthetic_y_true(BATCH, SIZE, CLASS_NUM, ANCHORS, rank, 'cuda')
    with y3.variable_scope('yolov3'):
        model.forward(torch.zeros((1, 64, 64, 3), device='cuda'))
        random_init(seed=1)                              # same weights on every rank
        upd = None
        if args.head_only:
            upd = [v for v in y3.global_variables(scope='yolov3') if v.op_name.startswith('yolov3/yolov3_head')]
        ws_arg = getattr(args, 'wgrad_stream', None)
        trainer = training.Trainer(model, config_optimizer('sgd', 1e-4), update_vars=upd,
                                   process_group=dist.group.WORLD if distributed and world > 1 else None,
                                   wgrad_stream='auto' if ws_arg is None else bool(ws_arg))
        # warm-up: the W asked for, and (default) the seven steps in which the Trainer measures its second stream against one
        # stream and settles on the faster (training.Trainer: wgrad_stream='auto'; the choice is reported below)
        done = 0
        while done < args.warmup or trainer.wgrad_choice is None:
            loss = trainer.step(x, yt)
            done += 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = trainer.step(x, yt)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        rank_ms = [round(v / args.steps * 1e3, 3) for v in getattr(max_over_ranks, 'per_rank', {}).get('last', [elapsed])]
        fw.check_context()
        loss0 = float(loss[0])
        assert np.isfinite(loss0), "non-finite loss"
        # the same steps with the batches coming from the feeder (decode + augmentation + resize + upload + target assignment
        # under the steps): never `value`, reported beside it
        fed = None
        if not args.no_fed:
            try:
                fed = fed_train_steps(args, trainer, torch, rank, barrier, max_over_ranks)
            except Exception as e:      # the fed repetition must never cost the resident measurement
                fed = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank != 0:
        return None
    table = [(l['k'], l['stride'], l['cin'], l['cout'], l['bn']) for l in model._train['topo'].layers]
    per_layer = conv_flops(table, BATCH, SIZE, SIZE)
    fwd_flops = float(per_layer.sum())
    # forward of every layer + data and weight gradients of the layers backward visits (head-only: the 23 head convs)
    bwd_flops = 2.0 * (float(per_layer[52:].sum()) if args.head_only else fwd_flops)
    ms = elapsed / args.steps * 1e3
    tflops = (fwd_flops + bwd_flops) / (ms * 1e-3) / 1e12
    # MFMA work the kernels ISSUE (same definition as the forward's roofline): a Winograd kernel issues 16/36 of the
    # direct count; forward and data gradient of every stride-1 3x3 conv, weight gradient where cin and cout are
    # multiples of 64 (y3_conv_wgrad_wino_eligible)
    wino = args.precision == 'f32_wino'
    s13 = np.array([wino and bool(engine.wino_eligible(k, s, ci, co)) for (k, s, ci, co, _) in table])
    wgw = s13 & np.array([ci % 64 == 0 and co % 64 == 0 for (_, _, ci, co, _) in table])
    visited = np.arange(len(table)) >= (52 if args.head_only else 0)
    # forward and data gradient of the layers y3_conv_wino44_preferred names run F(4x4,3x3): 36/144 (useful work; padding tiles
    # not counted); the data gradient is the conv with the channel axes swapped
    grids = layer_input_grids(table, SIZE)
    f_fwd = np.array([(36.0 / 144.0 if (bn and engine.wino44_preferred(BATCH, g, g, k, s, ci, co)) else 16.0 / 36.0) if w_ else 1.0
                      for (k, s, ci, co, bn), g, w_ in zip(table, grids, s13)])
    f_dg = np.array([(36.0 / 144.0 if engine.wino44_preferred(BATCH, g, g, k, s, co, ci) else 16.0 / 36.0) if w_ else 1.0
                     for (k, s, ci, co, bn), g, w_ in zip(table, grids, s13)])
    issued = (per_layer * f_fwd).sum() \
        + (per_layer * visited * f_dg).sum() \
        + (per_layer * visited * np.where(wgw, 16.0 / 36.0, 1.0)).sum()
    issued_tflops = float(issued) / (ms * 1e-3) / 1e12
    grad_bytes = int(trainer.flat.numel() * 4)
    return {
        "metric": "images/sec, train step at 416x416 bs=%d per GPU (forward + loss + backward + gradient all-reduce + clip + SGD)" % BATCH,
        "value": round(world * BATCH * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "per_rank_ms_per_step": rank_ms,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "precision": ("fp32 MFMA arithmetic; Winograd forms of the stride-1 3x3 convs - F(4x4,3x3) for the forward and the data "
                      "gradient where the launch fills the chip (Cin >= 64), F(2x2,3x3) elsewhere and for the weight "
                      "gradient -, direct kernels for the other layers" if args.precision == 'f32_wino'
                      else PRECISION_TEXT[args.precision]),
        "data": "synthetic",
        "config": {"workload": "configs[3]: train step, synthetic COCO-80 batches, 416x416 bs=%d per GPU, SGD, %s, "
                               "RCCL gradient all-reduce (bucketed, overlapped with backward)" %
                               (BATCH, "update_part = yolov3_head" if args.head_only else "whole model"),
                   "batch_per_gpu": BATCH, "global_batch": BATCH * world, "image_size": SIZE, "class_num": CLASS_NUM,
                   "parallelism": "dp%d (one all-reduce of %d bytes of fp32 gradients per step, %d buckets)" %
                                  (world, grad_bytes, len(trainer.exchange.edges))},
        "roofline": {"bound": "mfma", "achieved": round(issued_tflops, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(issued_tflops / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                     "achieved_algorithmic": round(tflops, 2),
                     "kernel": "whole train step (not one kernel): MFMA work ISSUED by forward + data gradient + weight "
                               "gradient of every layer backward visits (a Winograd kernel issues 16/36 of the direct "
                               "count) over the step time, BN / loss / update kernels included in the time; "
                               "achieved_algorithmic counts direct-convolution FLOPs"},
        "loss": round(loss0, 4), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 2),
        "wgrad_stream": {"on": bool(trainer.wgrad_choice), "how": "measured by the Trainer in its first steps" if ws_arg is None else "forced",
                         "calibration_ms": getattr(trainer, 'wgrad_calibration', None)},
        "fed": fed if fed is None or "error" in fed
        else dict(fed, images_per_s=round(world * BATCH / (fed["ms_per_step"] * 1e-3), 2)),
    }


def fed_train_steps(args, trainer, torch, rank, barrier, max_over_ranks):
    """The c4 step with its batches decoded, augmented ('train' chain with mix-up), resized, uploaded and target-assigned by
    the feeder while the device runs the previous step (reference train.py:34-58: num_parallel_calls = 10, prefetch 5)."""
    import importlib.util
    import pathlib
    import tempfile
    spec = importlib.util.spec_from_file_location('feeder_rate', os.path.join(ROOT, 'tools', 'feeder_rate.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    from yolov3_tensorflow_amd.feeder import Feeder
    warm, workers, prefetch = 2, 10, 5
    total = (args.steps + warm + 1) * BATCH
    with tempfile.TemporaryDirectory() as folder:
        lines = tool.write_set(pathlib.Path(folder), 256, seed=rank)
        lines = (lines * (total // len(lines) + 1))[:total]
        feeder = Feeder(lines, BATCH, CLASS_NUM, [SIZE, SIZE], tool.ANCHORS, mode='train', use_mix_up=True,
                        num_threads=workers, prefetch=prefetch, seed=1 + rank)
        it = feeder.epoch(0)
        for _ in range(warm):
            batch = next(it)
            loss = trainer.step(batch.images, batch.y_true)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            batch = next(it)
            loss = trainer.step(batch.images, batch.y_true)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        it.close()
        feeder.close()
    assert np.isfinite(float(loss[0])), "non-finite loss on fed batches"
    return {"ms_per_step": round(elapsed / args.steps * 1e3, 3), "steps": args.steps, "workers": workers,
            "prefetch": prefetch, "backend": feeder.backend, "pixels": feeder.pixels,
            "data": "synthetic 640x480 JPEGs through the reference's 'train' augmentation chain with mix-up"}


# ------------------------------------------------------------------------------------------------------------
# c2 / c5: the forward
# ------------------------------------------------------------------------------------------------------------
def run_forward(args, y3, torch, dist, rank, world, distributed, barrier, max_over_ranks):
    from yolov3_tensorflow_amd import engine, framework as fw
    bf16 = args.workload == 'c5'
    model = y3.yolov3(CLASS_NUM, ANCHORS)
    model.inference_streams = max(1, int(getattr(args, 'streams', 1)))
    model.compute_dtype = 'bf16' if bf16 else args.precision
    split = (not bf16) and args.precision in ('f32_bf16x6', 'f32_bf16x3')
    wino = (not bf16) and args.precision == 'f32_wino'
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
        if model.inference_streams > 1:      # keep two streams only where they measure faster on THIS box / process
            model.choose_inference_streams(x, candidates=(1, model.inference_streams))
        for i in range(args.warmup):
            # (two warm-up steps run with layer profiling on, so that the event objects the timed region records into
            # already exist: creating them inside the timed region stalled the host for milliseconds on a cold box)
            model.set_layer_profiling(i < 2)
            fms = model.forward(x, False)
        model.set_layer_profiling(False)
        if args.warmup:
            model.read_layer_ms()
        # per-layer hipEvents (for `roofline`) are recorded inside the timed region on every 8th step only: 76 event
        # records cost ~1.5 % of a profiled step
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            model.set_layer_profiling(i % 8 == 4)
            fms = model.forward(x, False)
        barrier()
        elapsed = time.perf_counter() - t0
        fw.check_context()      # a stream-K hand-off that timed out inside the timed region would raise here
        layer_ms, table = model.read_layer_ms()
        is_sk = model.layer_is_streamk(BATCH, SIZE, SIZE)
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

    # Secondary measurements (c2, default precision only; outside the timed region above and never `value`)
    fast, direct = None, None
    if not bf16 and not split and not args.no_secondary:
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
    elapsed = max_over_ranks(elapsed)
    rank_ms = [round(v / args.steps * 1e3, 4) for v in getattr(max_over_ranks, 'per_rank', {}).get('last', [elapsed])]
    if rank != 0:
        return None

    ms_per_step = elapsed / args.steps * 1e3
    value = world * BATCH * args.steps / elapsed
    flops = conv_flops(table, BATCH, SIZE, SIZE)
    fused = model.layer_fused(BATCH, SIZE, SIZE)
    nbytes = conv_bytes(table, BATCH, SIZE, SIZE, 2 if bf16 else 4, fused)
    is3 = np.array([k == 3 and cin != 3 for (k, s, cin, cout, bn) in table])
    is_wino, fac = winograd_issue_factors(table, args.precision if wino else 'f32', SIZE)
    issued = flops * fac                                           # MFMA work the kernels actually issue
    n_f44 = int(np.sum(is_wino & (fac < 16.0 / 36.0 - 1e-9)))      # layers on the F(4x4,3x3) kernel
    if wino:        # dominant family: the Winograd kernel (every stride-1 3x3 conv but the stem)
        dom = is_wino
    elif bf16:      # the 3x3 convs on 128x128 tiles (conv_mfma_bf16_kernel<128,128,2,2,3,false>)
        dom = np.array([k == 3 and cin != 3 and cout > 64 for (k, s, cin, cout, bn) in table])
    else:           # the stream-K 3x3 kernel
        dom = np.array(is_sk, bool)
    dom_ms = float(layer_ms[dom].sum())
    n_dom = int(dom.sum())
    achieved = float(issued[dom].sum()) / (dom_ms * 1e-3) / 1e12
    achieved_alg = float(flops[dom].sum()) / (dom_ms * 1e-3) / 1e12
    # split precisions: fp32-equivalent peak = dense bf16 MFMA peak / products per fp32 multiply-add
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else {'f32': PEAK_FP32_MFMA_TFLOPS, 'f32_wino': PEAK_FP32_MFMA_TFLOPS,
                                                'f32_bf16x6': PEAK_BF16_MFMA_TFLOPS / 6,
                                                'f32_bf16x3': PEAK_BF16_MFMA_TFLOPS / 3}[args.precision]
    bound_ms = np.maximum(nbytes / (PEAK_HBM_TBPS * 1e12), issued / (peak * 1e12)) * 1e3      # per layer
    traffic, traffic_src = (None, None) if split else traffic_from_profile(
        ['r06_pmc_traffic_bf16.json', 'r05_pmc_traffic_bf16.json', 'r04_pmc_traffic_bf16.json', 'r03_pmc_traffic_bf16.json'] if bf16 else
        ['r06_pmc_traffic_wino.json', 'r05_pmc_traffic_wino.json', 'r04_pmc_traffic_wino.json', 'r03_pmc_traffic_wino.json', 'r02_pmc_traffic_wino.json'] if wino else ['r03_pmc_traffic.json', 'r01_pmc_traffic.json'])
    kernel = ("conv_bf16p_kernel / conv_bf16x_kernel (3x3 implicit-GEMM convs with Cout > 64, bf16 storage, LDS-DMA staged: "
              "the pipelined kernel on 192x256 / 192x128 tiles where the library's tile model picks them - the 76-, 38- and "
              "19-grid layers at this size -, 256-row and 128x128 tiles elsewhere; see DESIGN.md 4.3)" if bf16 else
              "conv_mfma_split_kernel<128,128,2,2,3,false,true,%d,false> (3x3 implicit-GEMM conv on the bf16 matrix pipe, "
              "stream-K schedule; peak = 2500/%d fp32-equivalent)" % ((3, 6) if args.precision == 'f32_bf16x6' else (2, 3))
              if split else
              "the Winograd kernels of the stride-1 3x3 convs: conv_wino8_f32_kernel (F(2x2,3x3), %d launches) and the "
              "F(4x4,3x3) form (%d layers: the convs with Cin >= 64) - one kernel (conv_wino44_f32_kernel, input transform "
              "inside the K-loop) where Cout < 512, two kernels (wino44_input_transform_kernel writes V = B^T d B once, "
              "conv_wino44v_f32_kernel runs the 36 batched GEMMs + the output transform) where Cout >= 512; a layer's time is "
              "the time of ALL its kernels (hipEvents around the layer); `achieved` = useful MFMA work "
              "per second (16/36 resp. 36/144 of the direct-convolution FLOPs, per layer by the kernel the library picked; "
              "zero tiles padding a 13-/26-grid to a multiple of 4 are issued but not counted); `achieved_algorithmic` "
              "counts the direct-convolution FLOPs" % (int(is_wino.sum()) - n_f44, n_f44)
              if wino else
              "conv_mfma_f32_kernel<128,128,2,2,3,false,true,false> (3x3 implicit-GEMM conv, stream-K schedule)")
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
                   "class_num": CLASS_NUM, "parallelism": "replicas (image-sharded, no collective)",
                   "streams_per_gpu": model.inference_streams,
                   "streams_calibration_ms": {str(k): round(v, 3) for k, v in
                                              getattr(model, 'inference_streams_timing', {}).items()} or None,
                   "streams_note": ("the batch runs as %d equal parts on %d HIP streams of the GPU (north star: independent per-GPU "
                                    "streams for inference; one part's kernel tails and partly filled rounds of workgroups are "
                                    "filled by the other's kernels: A/B in one process 10.87 -> 10.33 ms for c2 (round 4, tools/streams_ab.py; "
                                    "c5's round-5 tiles fill the CUs with the whole batch and usually keep one stream); kept only because choose_inference_streams() measured it faster "
                                    "than one stream in THIS process before the warm-up); the forwards that carry per-layer hipEvents (every 8th step of "
                                    "the timed region: the `roofline` figures) run on ONE stream, so kernel durations are those "
                                    "of undisturbed launches of the whole batch" % (model.inference_streams, model.inference_streams))
                                   if model.inference_streams > 1 else "one stream"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 2),
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": kernel,
                     "launches_per_step": n_dom,
                     "avg_launch_ms": round(dom_ms / n_dom, 4),
                     "issued_gflop_per_launch": round(float(issued[dom].sum()) / n_dom / 1e9, 3),
                     "algorithmic_gflop_per_launch": round(float(flops[dom].sum()) / n_dom / 1e9, 3),
                     "algorithmic_mb_per_launch": round(float(nbytes[dom].sum()) / n_dom / 1e6, 2),
                     "achieved_algorithmic": round(achieved_alg, 2),
                     "all_3x3_tflops": round(float(flops[is3].sum()) / (float(layer_ms[is3].sum()) * 1e-3) / 1e12, 2),
                     "whole_forward_tflops": round(float(flops.sum()) / (ms_per_step * 1e-3) / 1e12, 2),
                     "whole_forward_issued_tflops": round(float(issued.sum()) / (ms_per_step * 1e-3) / 1e12, 2),
                     "whole_forward_bound_ms": round(float(bound_ms.sum()), 4),
                     "whole_forward_frac": round(float(bound_ms.sum()) / ms_per_step, 4),
                     "whole_forward_hbm_tbps": round(float(nbytes.sum()) / (ms_per_step * 1e-3) / 1e12, 3),
                     "sum_layer_ms": round(float(layer_ms.sum()), 4)},
    }
    if direct is not None:
        out["direct_path"] = direct
    if fast is not None:
        out["fast_path"] = fast
    out["per_rank_ms_per_step"] = rank_ms          # every rank's own timed region (value uses the slowest)
    if world == 1 and not bf16 and not getattr(args, 'secondary', False):      # checker work: N = 1 only
        try:
            out["box_delta_vs_oracle"] = box_delta_vs_oracle(model, y3, x, fms)
        except Exception as e:      # a checker must never cost the line
            out["box_delta_vs_oracle"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1 and not args.no_cpu_baseline and not bf16:
        out["cpu_baseline"] = cpu_baseline(y3.global_variables(scope='yolov3'))
    return out


if __name__ == '__main__':
    main()
