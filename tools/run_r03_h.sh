#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_bench_config_train_gpu.py tests/test_data_parallel_gpu.py tests/test_rccl_gpu.py -m gpu -x -q -s > $O/tests_train.log 2>&1; echo "train tests rc=$?" | tee -a $O/summary.txt
grep -E "bs=64|bs=8 @416|gradient rel err|passed|failed|Error|error" $O/tests_train.log | tail -14
timeout 600 python -m pytest tests/test_feeder_gpu.py tests/test_train_script_gpu.py -m gpu -x -q -s > $O/tests_feeder.log 2>&1; echo "feeder/train-script tests rc=$?" | tee -a $O/summary.txt
grep -E "train step bs|passed|failed|Error|error" $O/tests_feeder.log | tail -6
timeout 300 python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" | tee -a $O/summary.txt
python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['frac'], 'mem', d['peak_mem_gb'])" | tee -a $O/summary.txt
