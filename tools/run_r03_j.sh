#!/bin/bash
# round 3, step j: packed-FMA stem vs matrix-pipe stem; the lane-per-record loss kernel (parity + c4 bench)
O=gpurun_out/r03j; mkdir -p $O
for cfg in "Y3_STEM_PK=1" "Y3_STEM_MFMA=1"; do
  env $cfg timeout 300 python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20 --csv $O/layers_c5_${cfg//=/_}.csv > $O/layers_c5_${cfg//=/_}.txt 2>&1
  echo "c5 $cfg: stem $(sed -n 2p $O/layers_c5_${cfg//=/_}.csv) | $(tail -2 $O/layers_c5_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
  env $cfg timeout 300 python tools/layer_profile.py --batch 32 --size 416 --precision f32_wino --iters 20 --csv $O/layers_c2_${cfg//=/_}.csv > $O/layers_c2_${cfg//=/_}.txt 2>&1
  echo "c2 $cfg: stem $(sed -n 2p $O/layers_c2_${cfg//=/_}.csv) | $(tail -2 $O/layers_c2_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
done
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_bench_config_train_gpu.py tests/test_conv_gpu.py tests/test_bf16_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "loss/train/conv/bf16 tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
timeout 300 python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" | tee -a $O/summary.txt
python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['frac'])" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --no-cpu-baseline --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_c4.log 2>&1; echo "rocprof c4 rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
find $GRAFT_REPO_ROOT/$O/prof_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/c4_kernel_stats.csv
find $GRAFT_REPO_ROOT/$O/prof_c4 -type f ! -name "*stats.csv" -delete 2>/dev/null
