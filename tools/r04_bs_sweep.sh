R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bs_sweep; mkdir -p $O; cd $R
export Y3_LIB_PATH=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so
for bs in 4 8 16; do
  for m in 0 1 2; do
    Y3_WINO44=$m timeout 120 python tools/layer_profile.py --batch $bs --precision f32_wino --csv $O/l_${bs}_$m.csv 2>&1 | grep "^total" | sed "s/^/bs=$bs WINO44=$m /"
  done
done
