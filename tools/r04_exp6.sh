R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/exp6; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_forward_gpu.py tests/test_bench_config_gpu.py -x -q 2>&1 | tail -2
export Y3_LIB_PATH=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so
for i in 1 2; do
python tools/layer_profile.py --precision f32_wino --csv $O/layers_res_$i.csv > $O/l.log 2>&1; tail -3 $O/l.log | head -2
Y3_CONV_RESIDENT_OFF=1 python tools/layer_profile.py --precision f32_wino --csv $O/layers_off_$i.csv > $O/l.log 2>&1; tail -3 $O/l.log | head -2
done
