R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/exp4; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_bench_config_gpu.py tests/test_forward_gpu.py -x -q 2>&1 | tail -3
Y3_WINO44=2 python tools/layer_profile.py --precision f32_wino --csv $O/layers_all44.csv > $O/layers_all44.log 2>&1; tail -4 $O/layers_all44.log
python tools/layer_profile.py --precision f32_wino --csv $O/layers_def.csv > $O/layers_def.log 2>&1; tail -4 $O/layers_def.log
Y3_WINO44=2 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-400
