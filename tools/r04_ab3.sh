# A/B/C of libraries inside ONE call: whole-forward layer profile (tools/layer_profile.py, fp32 Winograd mode), alternating
# usage: r04_ab3.sh <reps> <lib> [<lib> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab3; mkdir -p $O; cd $R
N=$1; shift
for i in $(seq 1 $N); do
  for L in "$@"; do
    Y3_LIB_PATH=$L timeout 120 python tools/layer_profile.py --precision f32_wino --csv $O/$(basename $L .so)_$i.csv 2>&1 | grep -E "^total" | tr '\n' ' '; echo " <- $(basename $L)"
  done
done
