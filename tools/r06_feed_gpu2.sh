#!/bin/bash
# round 6: device pixels as the Feeder's default - feeder tests, kernel times of a bs=64 batch, the fed c4 step
set -x
cd /root/repo
timeout 1500 python -m pytest tests/test_feed_gpu.py tests/test_feeder_gpu.py -q -x 2>&1 | tail -15 > gpurun_out/r06_feed_gpu_tests2.txt
cat gpurun_out/r06_feed_gpu_tests2.txt
export TMPDIR=/tmp
rm -rf gpurun_out/feedprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/feedprof -- python tools/feeder_rate.py --workers 10 --backends thread --native 1 --pixels gpu --batches 16 > gpurun_out/r06_feed_prof_run.txt 2>&1
f=$(find gpurun_out/feedprof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r06_feed_kernel_stats.csv
head -8 gpurun_out/r06_feed_kernel_stats.csv
rm -rf gpurun_out/feedprof
timeout 900 python bench.py --workload c4 --steps 8 --warmup 2 > gpurun_out/r06_bench_c4_fed_gpu_pixels.json 2> gpurun_out/r06_bench_c4_fed_err.txt
cat gpurun_out/r06_bench_c4_fed_gpu_pixels.json
timeout 600 python bench.py --workload feeder --steps 16 --warmup 4 > gpurun_out/r06_bench_feeder.json 2>> gpurun_out/r06_bench_c4_fed_err.txt
cat gpurun_out/r06_bench_feeder.json
tail -5 gpurun_out/r06_bench_c4_fed_err.txt
