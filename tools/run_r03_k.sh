#!/bin/bash
# round 3, step k: stem with scalar-loaded weights (default) vs the matrix-pipe stem; conv parity; c4 kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
for cfg in "Y3_STEM_SGPR=1" "Y3_STEM_MFMA=1"; do
  env $cfg timeout 300 python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20 --csv $O/layers_c5_${cfg//=/_}.csv > $O/layers_c5_${cfg//=/_}.txt 2>&1
  echo "c5 $cfg: stem $(sed -n 2p $O/layers_c5_${cfg//=/_}.csv) | $(tail -2 $O/layers_c5_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
  env $cfg timeout 300 python tools/layer_profile.py --batch 32 --size 416 --precision f32_wino --iters 20 --csv $O/layers_c2_${cfg//=/_}.csv > $O/layers_c2_${cfg//=/_}.txt 2>&1
  echo "c2 $cfg: stem $(sed -n 2p $O/layers_c2_${cfg//=/_}.csv) | $(tail -2 $O/layers_c2_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_bf16_gpu.py tests/test_forward_gpu.py tests/test_bench_config_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "conv/bf16/forward tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python $R/bench.py --workload c4 --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_c4.log 2>&1; echo "rocprof c4 rc=$?" | tee -a $O/summary.txt
rm -f $O/prof*/p_kernel_trace.csv 2>/dev/null
head -12 $O/prof_c4/p_kernel_stats.csv | cut -c1-160
