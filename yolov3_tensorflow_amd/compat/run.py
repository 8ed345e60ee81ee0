# coding: utf-8
"""python -m yolov3_tensorflow_amd.compat.run <script.py> [script arguments...]  (see the package docstring)."""
import os
import sys

from . import install


def run_script(argv):
    """Execute `argv[0]` as __main__ with the shims installed and sys.argv = argv; returns the script's globals (what a
    harness inspects afterwards: the reference's scripts keep their results in module-level names)."""
    script = argv[0]
    install()
    # the script's own directory must NOT shadow the shims (the reference keeps its model.py / utils/ next to its
    # scripts): the script is exec'ed here instead of going through runpy, which would put that directory first
    sys.argv = [script] + argv[1:]
    script_dir = os.path.dirname(os.path.abspath(script))
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or '.') != script_dir or p == '']
    # ... but its sibling modules that have no shim (the reference's `args.py`, ref: train.py:10) must still resolve:
    # the directory goes LAST, so that model / utils / tensorflow / cv2 keep resolving to the shims
    sys.path.append(script_dir)
    code = compile(open(script, 'rb').read(), script, 'exec')
    glob = {'__name__': '__main__', '__file__': script, '__builtins__': __builtins__}
    exec(code, glob)
    return glob


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        sys.stderr.write(__doc__ + "\n")
        return 2
    run_script(argv)
    return 0


if __name__ == '__main__':
    sys.exit(main())
