#!/bin/bash
# Knock-out / variant builds of conv_wino44v_f32_kernel (tools/build_variant.py ... -DW44V_KO_*), the two-kernel F(4x4) form only,
# all inside ONE gpurun call (boxes differ by 3-5 %).  Output: gpurun_out/r06_w44_ko.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_w44_ko.txt
: > $O
cd /tmp && export TMPDIR=/tmp
for v in product ko_b ko_dma ko_a ko_tail ko_all ko_all_tail bd6 ad2 product; do
  if [ $v = product ]; then unset Y3_LIB_PATH; else export Y3_LIB_PATH=$R/tools/_probe/lib_$v.so; fi
  echo "== $v" >> $O
  rm -rf /tmp/prof_$v
  W44_ONLY2K=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -o t -- python $R/tools/wino44_bench.py 32 > /dev/null 2>&1
  python $R/tools/trace_by_grid.py /tmp/prof_$v/t_kernel_trace.csv "wino44v|input_transform" >> $O
done
cat $O
