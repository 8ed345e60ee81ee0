"""The conv code paths that sit behind environment switches of the EXPERIMENTS build of the library
(csrc/libyolo355_exp.so; the product library reads no environment variable; the switches are read once per process, so
each case runs in its own interpreter): the hybrid stream-K schedule of the F(2x2,3x3) kernel (Y3_WINO_SK_HYBRID=1) and
the data-parallel schedules forced where the product picks stream-K (Y3_CONV_WINO_STREAMK=0 + Y3_CONV_STREAMK=0), each
against the fp64 reference through the same cases as the default path (tests/test_conv_gpu.py), plus the statistics
epilogue and the data / weight gradients of the train step; and every tile shape of the bf16 kernels forced in turn
(Y3_BF16X_TILE=A..E for the 3x3 convs; Y3_BF16R=1 sends every 1x1 conv with Cin % 64 == 0 to the ring kernel - the product
only those with Cin >= 512 - with Y3_BF16R_TILE=a..g forcing its tile; Y3_BF16R=0 none) through the bf16 conv cases of tests/test_bf16_gpu.py
(the every-Cin ring setting also through the whole bf16 forward: the fused first residual block reads another weight packing
and must step aside); and each form of the F(4x4,3x3) conv forced on EVERY case (Y3_WINO44_V=1 two kernels, =0 one kernel -
the product takes two where Cout >= 512), the whole fp32 forward included."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('env', [
    {'Y3_WINO_SK_HYBRID': '1'},
    {'Y3_CONV_WINO_STREAMK': '0', 'Y3_CONV_STREAMK': '0'},
], ids=['wino_hybrid_schedule', 'data_parallel_schedules'])
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


@pytest.mark.parametrize('env', [{'Y3_BF16X_TILE': t} for t in 'ABCDE'] +
                         [{'Y3_BF16R': '1', 'Y3_BF16R_TILE': t} for t in 'abcdefg'] + [{'Y3_BF16R': '1'}, {'Y3_BF16R': '0'}],
                         ids=['3x3_tile_' + t for t in 'ABCDE'] + ['1x1_ring_tile_' + t for t in 'abcdefg'] +
                             ['1x1_ring_every_cin', '1x1_register_staged'])
def test_forced_bf16_tiles(env):
    from yolov3_tensorflow_amd import build
    e = dict(os.environ)
    e.update(env)
    e['Y3_LIB_PATH'] = build.build_experiments(verbose=False)
    what = [os.path.join(HERE, 'test_bf16_gpu.py') + '::test_bf16_conv_matches_fp64_on_rounded_operands']
    if env == {'Y3_BF16R': '1'}:      # ... and the whole forward: every 1x1 conv of the net on the ring kernel's packing
        what.append(os.path.join(HERE, 'test_bf16_gpu.py') + '::test_bf16_forward_tracks_the_fp32_oracle')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu'] + what,
                       env=e, cwd=os.path.dirname(HERE), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0, out[-3000:]
    assert ' passed' in out, out[-1000:]


@pytest.mark.parametrize('env', [{'Y3_WINO44_V': '1'}, {'Y3_WINO44_V': '0'}], ids=['f4x4_two_kernels_everywhere', 'f4x4_one_kernel_everywhere'])
def test_forced_f4x4_forms(env):
    from yolov3_tensorflow_amd import build
    e = dict(os.environ)
    e.update(env)
    e['Y3_LIB_PATH'] = build.build_experiments(verbose=False)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu',
                        os.path.join(HERE, 'test_conv_gpu.py') + '::test_winograd_f4x4_conv_matches_fp64',
                        os.path.join(HERE, 'test_train_gpu.py') + '::test_winograd_f4x4_data_gradient_matches_autograd',
                        os.path.join(HERE, 'test_train_gpu.py') + '::test_conv_epilogue_statistics_equal_the_separate_pass',
                        os.path.join(HERE, 'test_forward_gpu.py') + '::test_forward_matches_oracle_416'],
                       env=e, cwd=os.path.dirname(HERE), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0, out[-3000:]
    assert ' passed' in out, out[-1000:]
