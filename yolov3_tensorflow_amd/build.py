"""Build libyolo355.so (the C-ABI HIP library) in-tree for gfx950.

    python -m yolov3_tensorflow_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so travels with the repo snapshot to the GPU box.
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
    ("y3_conv.hip", []),
    ("y3_conv_bf16.hip", []),
    ("y3_conv_split.hip", []),
    ("y3_conv_wino.hip", []),
    ("y3_decode.hip", ["-ffp-contract=off"]),
    ("y3_nms.hip", ["-ffp-contract=off"]),
    ("y3_ops.hip", ["-ffp-contract=off"]),
    ("y3_train.hip", ["-ffp-contract=off"]),
    ("y3_wgrad.hip", []),
    ("y3_wgrad_wino.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("Y3_EXTRA_HIPCC_FLAGS", "").split()   # experiment hook (e.g. -DY3_EXP=1)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s, _ in SOURCES] + [
        os.path.join(CSRC, "y3_internal.h"),
        os.path.join(CSRC, "y3_conv_common.h"),
        os.path.join(HERE, "..", "include", "yolo355.h"),
        os.path.abspath(__file__),
    ]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
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
