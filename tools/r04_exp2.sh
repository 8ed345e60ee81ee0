R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/exp2; mkdir -p $O; cd $R
for v in "" KO_DMA KO_B KO_TRANSFORM KO_AFRAG KO_DMA_B KO_ALL; do
  echo "== variant ${v:-product}"
  if [ -z "$v" ]; then python tools/wino44_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_product.txt
  else Y3_LIB_PATH=$R/tools/_probe/lib_$v.so python tools/wino44_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_$v.txt; fi
done
