#!/bin/bash
# direct conv kernel A/B (product against tools/_probe/lib_conv_old.so) inside ONE gpurun call: conv_bench rows and the c2 forward
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_conv_ab.txt
: > $O
for rep in 1 2; do
for v in product conv_old; do
  if [ $v = product ]; then unset Y3_LIB_PATH; else export Y3_LIB_PATH=$R/tools/_probe/lib_$v.so; fi
  echo "== $v" >> $O
  for row in 6 7 8 9 10 11 12 13; do python $R/tools/conv_bench.py --only $row 2>&1 | grep "H=" >> $O; done
  line=$(python $R/bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  echo "c2 one stream: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.1f ms %.3f" % (d["value"], d["ms_per_step"]))')" >> $O
done; done
cat $O
