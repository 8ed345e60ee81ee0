# A/B of two libraries inside ONE call (boxes differ by several per cent): whole-forward layer profile, alternating
# usage: r04_ab.sh <libA> <libB> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    Y3_LIB_PATH=$L python tools/layer_profile.py --precision f32_wino --csv $O/$(basename $L .so)_$i.csv 2>&1 | grep -E "^total|^k=3" | tr '\n' ' '; echo " <- $(basename $L)"
  done
done
