"""Build libyolo355.so (the C-ABI HIP library, gfx950) and liby3feed.so (the feeder's host-side image library) in-tree.

    python -m yolov3_tensorflow_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so files travel with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libyolo355.so")

# (source, extra flags).  NMS and decode are compiled without FMA contraction so that their fp32
# arithmetic follows the reference's operation order exactly (bit-exact NMS decisions).
SOURCES = [
    ("y3_abi.hip", []),
    ("y3_net_train.hip", []),
    ("y3_conv.hip", []),
    ("y3_conv_f32s.hip", []),
    ("y3_conv_bf16.hip", []),
    ("y3_conv_bf16x.hip", []),
    ("y3_conv_bf16r.hip", []),
    ("y3_conv_bf16s.hip", []),
    ("y3_conv_bf16b.hip", []),
    ("y3_conv_split.hip", []),
    ("y3_conv_wino.hip", []),
    ("y3_conv_wino44.hip", []),
    ("y3_decode.hip", ["-ffp-contract=off"]),
    ("y3_nms.hip", ["-ffp-contract=off"]),
    ("y3_ops.hip", ["-ffp-contract=off"]),
    ("y3_train.hip", ["-ffp-contract=off"]),
    ("y3_wgrad.hip", []),
    ("y3_wgrad_wino.hip", []),
    ("y3_feed_gpu.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


# liby3feed.so: host code (include/yolo355_feed.h).  No FMA contraction and no fast-math: it reproduces numpy's float32
# loops and Pillow's C arithmetic bit for bit.  Baseline x86-64 only (the file is built here and runs on the GPU box).
FEED_SRC = os.path.join(CSRC, "y3_feed.cpp")
FEED_LIB = os.path.join(CSRC, "liby3feed.so")
FEED_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-Wall",
              "-Wno-format-truncation"]


def build_feed(force=False, verbose=True):
    deps = [FEED_SRC, os.path.join(HERE, "..", "include", "yolo355_feed.h")]
    if not force and os.path.exists(FEED_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(FEED_LIB) for d in deps):
        return FEED_LIB
    tmp = FEED_LIB + ".%d.tmp" % os.getpid()         # (several feeder workers may get here at once: rename is atomic)
    cmd = [os.environ.get("CXX", "g++")] + FEED_FLAGS + [FEED_SRC, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError("building liby3feed.so failed:\n%s\n%s" % (" ".join(cmd), out.stdout.decode(errors="replace")))
    os.replace(tmp, FEED_LIB)
    return FEED_LIB


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def csrc_sha16():
    """Hash of the kernel sources (csrc/*.hip, csrc/*.h, include/yolo355.h): stamps the PMC traffic files under
    profiles/ so that bench.py can tell a figure taken on THIS build from a stale one."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(n for n in os.listdir(CSRC) if n.endswith(('.hip', '.h')))
    for path in [os.path.join(CSRC, n) for n in names] + [os.path.join(HERE, "..", "include", "yolo355.h")]:
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s, _ in SOURCES] + [
        os.path.join(CSRC, "y3_internal.h"),
        os.path.join(CSRC, "y3_conv_common.h"),
        os.path.join(CSRC, "y3_net.h"),
        os.path.join(CSRC, "y3_feed_px.h"),
        os.path.join(HERE, "..", "include", "yolo355_feed.h"),
        os.path.join(HERE, "..", "include", "yolo355.h"),
        os.path.abspath(__file__),
    ]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# The sources with experiment switches (y3_exp_env, csrc/y3_internal.h): compiled a second time with -DY3_EXPERIMENTS
# into libyolo355_exp.so, the library tools/ A/B runs and tests/test_conv_variants_gpu.py select with Y3_LIB_PATH.
# The product library reads no environment variable.
EXP_LIB = os.path.join(CSRC, "libyolo355_exp.so")
EXP_SOURCES = ("y3_conv.hip", "y3_conv_f32s.hip", "y3_conv_bf16x.hip", "y3_conv_bf16r.hip", "y3_conv_bf16s.hip", "y3_conv_bf16b.hip", "y3_conv_split.hip", "y3_conv_wino.hip", "y3_conv_wino44.hip",
               "y3_wgrad.hip")


def build_experiments(verbose=True):
    """libyolo355_exp.so = the product objects, with EXP_SOURCES recompiled under -DY3_EXPERIMENTS."""
    build(verbose=verbose)
    deps = [os.path.join(CSRC, s) for s in EXP_SOURCES] + [LIB]
    if os.path.exists(EXP_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(EXP_LIB) for d in deps):
        return EXP_LIB
    hipcc = _hipcc()
    procs, objs = [], []
    for src, extra in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if src in EXP_SOURCES:
            obj = os.path.join(CSRC, src.replace(".hip", ".exp.o"))
            cmd = [hipcc] + COMMON + extra + ["-DY3_EXPERIMENTS", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", EXP_LIB] + objs)
    return EXP_LIB


def build(force=False, verbose=True):
    build_feed(force=force, verbose=verbose)
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src, extra in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + COMMON + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
        if verbose and out.strip():
            print(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    if "--experiments" in sys.argv:
        print(build_experiments())
