R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/exp3; mkdir -p $O; cd $R
for v in "" $@; do
  echo "== variant ${v:-product}"
  if [ -n "$v" ]; then export Y3_LIB_PATH=$R/tools/_probe/lib_$v.so; fi
  true
  python tools/wino44_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_${v:-product}.txt
done
