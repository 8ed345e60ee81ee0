#!/bin/bash
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -s > $O/tests_bf16.log 2>&1; echo "bf16 tests rc=$?" | tee -a $O/summary.txt
grep -E "bs=16|passed|failed|Error|error|assert" $O/tests_bf16.log | tail -8
for cfg in "Y3_BF16X=1" "Y3_BF16X=0"; do
  env $cfg timeout 300 python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20 --csv $O/layers_${cfg//=/_}.csv > $O/layers_${cfg//=/_}.txt 2>&1
  echo "$cfg: $(tail -3 $O/layers_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
done
timeout 300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?" | tee -a $O/summary.txt
python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('c5:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['whole_forward_frac'])" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_feeder_gpu.py tests/test_train_script_gpu.py -m gpu -x -q -s > $O/tests_feeder.log 2>&1; echo "feeder + train script tests rc=$?" | tee -a $O/summary.txt
grep -E "train step bs|passed|failed|Error|error|assert" $O/tests_feeder.log | tail -8
