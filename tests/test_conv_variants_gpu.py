"""The Winograd code paths that sit behind environment switches of the EXPERIMENTS build of the library
(csrc/libyolo355_exp.so; the product library reads no environment variable; read once per process, so each runs
in its own interpreter): the four-wave kernel (Y3_WINO8=0, the round-1/2 kernel, still shipped) and the hybrid
stream-K schedule (Y3_WINO_SK_HYBRID=1), each against the fp64 reference and the direct kernel through the same
cases as the default path (tests/test_conv_gpu.py::test_winograd_conv_matches_fp64), plus the statistics epilogue and
the Winograd data gradient of the train step."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('env', [
    {'Y3_WINO8': '0'},
    {'Y3_WINO8': '0', 'Y3_WINO_SK_HYBRID': '1'},
    {'Y3_WINO8': '1', 'Y3_WINO_SK_HYBRID': '1'},
], ids=['four_wave', 'four_wave_hybrid', 'eight_wave_hybrid'])
def test_switched_winograd_paths(env):
    from yolov3_tensorflow_amd import build
    e = dict(os.environ)
    e.update(env)
    e['Y3_LIB_PATH'] = build.build_experiments(verbose=False)      # the switches exist only in the -DY3_EXPERIMENTS library
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu',
                        os.path.join(HERE, 'test_conv_gpu.py') + '::test_winograd_conv_matches_fp64',
                        os.path.join(HERE, 'test_train_gpu.py'), '-k',
                        'winograd_conv_matches_fp64 or conv_epilogue_statistics or conv_wgrad_and_dgrad'],
                       env=e, cwd=os.path.dirname(HERE), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0, out[-3000:]
    assert ' passed' in out, out[-1000:]
