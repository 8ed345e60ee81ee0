#!/usr/bin/env python
"""Variant build of libyolo355.so for A/B runs on the GPU box: recompile ONE source with extra -D flags, link it with the
product build's other objects into tools/_probe/lib_<name>.so (git-ignored; it travels with the gpurun snapshot) and
select it with Y3_LIB_PATH.

    python tools/build_variant.py <name> <source.hip> [-DFLAG ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from yolov3_tensorflow_amd import build as b
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    b.build(verbose=False)                       # the product objects are current
    out_dir = os.path.join(ROOT, 'tools', '_probe')
    os.makedirs(out_dir, exist_ok=True)
    extra = dict(b.SOURCES)[src]
    obj = os.path.join(out_dir, '%s_%s.o' % (name, src.replace('.hip', '')))
    cmd = [b._hipcc()] + b.COMMON + extra + flags + ['-c', os.path.join(b.CSRC, src), '-o', obj]
    subprocess.check_call(cmd)
    objs = [obj if s == src else os.path.join(b.CSRC, s.replace('.hip', '.o')) for s, _ in b.SOURCES]
    lib = os.path.join(out_dir, 'lib_%s.so' % name)
    subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    print(lib)


if __name__ == '__main__':
    main()
