#!/bin/bash
# the direct conv kernels' epilogue (loads first, one wait, stores back to back) against the build before it
# (tools/_probe/lib_old.so), inside ONE gpurun call: parity tests first, then conv_bench rows, the c2 forward, the c4 step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_epi_ab.txt
cd $R
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_train_gpu.py -q -x 2>&1 | tail -8 > $R/gpurun_out/r06_epi_tests.txt
cat $R/gpurun_out/r06_epi_tests.txt
: > $O
for rep in 1 2; do
for v in product old; do
  if [ $v = product ]; then unset Y3_LIB_PATH; else export Y3_LIB_PATH=$R/tools/_probe/lib_$v.so; fi
  echo "== $v" >> $O
  for row in 6 7 8 9 11 12 13; do python $R/tools/conv_bench.py --only $row 2>&1 | grep "H=" >> $O; done
  line=$(python $R/bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  echo "c2 one stream: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.1f ms %.3f" % (d["value"], d["ms_per_step"]))')" >> $O
  line=$(python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  echo "c2 two streams: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.1f ms %.3f" % (d["value"], d["ms_per_step"]))')" >> $O
  line=$(python $R/bench.py --workload c4 --steps 6 --warmup 2 --no-fed 2>/dev/null | tail -1)
  echo "c4: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms %.3f" % (d["ms_per_step"]))')" >> $O
done; done
cat $O
