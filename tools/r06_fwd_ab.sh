#!/bin/bash
# c2 forward A/B inside ONE gpurun call: product library against experiment-library settings (one stream and two).
# Output: gpurun_out/r06_fwd_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_fwd_ab.txt
: > $O
run() {  # label, env...
  label=$1; shift
  for streams in 1 2; do
    line=$(env "$@" python $R/bench.py --streams $streams --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
    echo "$label streams=$streams $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.1f ms %.3f frac %s" % (d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac")))')" | tee -a $O
  done
}
run product A=1
run exp_default Y3_LIB_PATH=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so
run exp_all_onekernel Y3_LIB_PATH=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so Y3_WINO44_V=0
run exp_all_twokernel Y3_LIB_PATH=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so Y3_WINO44_V=1
run exp_wino44_every Y3_LIB_PATH=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so Y3_WINO44=2
run product A=1
