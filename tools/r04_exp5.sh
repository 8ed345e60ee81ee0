R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/exp5; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q -k "f4x4 or wino" 2>&1 | tail -2
python tools/wino44_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
for v in $@; do echo "== $v"; Y3_LIB_PATH=$R/tools/_probe/lib_$v.so python tools/wino44_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_$v.txt; done
Y3_LIB_PATH=tools/_probe/lib_probe.so python tools/wino44_probe.py 32 52 128 256 2>&1 | grep -v amdgpu.ids
Y3_LIB_PATH=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so Y3_WINO44=2 python tools/layer_profile.py --precision f32_wino --csv $O/layers_all44.csv > $O/layers_all44.log 2>&1; tail -4 $O/layers_all44.log
