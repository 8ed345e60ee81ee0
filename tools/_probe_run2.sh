echo "tag            13x512->1024 26x256->512 52x128->256 104x64->128 208x32->64  (us per launch)"
Y3_TAG=wino4 Y3_WINO8=0 python tools/wino8_probe.py 2>&1 | grep -v amdgpu.ids
Y3_TAG=wino8 python tools/wino8_probe.py 2>&1 | grep -v amdgpu.ids
Y3_TAG=wino4_nores Y3_PROBE_NORES=1 Y3_WINO8=0 python tools/wino8_probe.py 2>&1 | grep -v amdgpu.ids
Y3_TAG=wino8_nores Y3_PROBE_NORES=1 python tools/wino8_probe.py 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_conv_gpu.py -x -q -m gpu 2>&1 | tail -1
