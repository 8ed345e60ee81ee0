# A/B of two libraries inside ONE call, both precisions: whole-forward layer profile (tools/layer_profile.py), alternating
# usage: r04_ab2.sh <libA> <libB> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab2; mkdir -p $O; cd $R
A=$1; B=$2; N=${3:-2}
for P in f32_wino bf16; do
 for i in $(seq 1 $N); do
  for L in $A $B; do
    Y3_LIB_PATH=$L timeout 120 python tools/layer_profile.py --precision $P --csv $O/${P}_$(basename $L .so)_$i.csv 2>&1 | grep -E "^total" | tr '\n' ' '; echo " <- $P $(basename $L)"
  done
 done
done
