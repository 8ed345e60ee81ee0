#!/bin/bash
# c5 (608x608 bf16 bs=16) A/B inside ONE gpurun call: product against a variant library.  Output: gpurun_out/r06_c5_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_c5_ab.txt
: > $O
for rep in 1 2; do
for v in product bf16x_old; do
  if [ $v = product ]; then unset Y3_LIB_PATH; else export Y3_LIB_PATH=$R/tools/_probe/lib_$v.so; fi
  line=$(python $R/bench.py --workload c5 --streams 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  echo "$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.1f ms %.3f frac %s" % (d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac")))')" | tee -a $O
done; done
