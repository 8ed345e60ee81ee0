#!/bin/bash
# round 6: the feeder's pixel work on the device - parity tests, then rates and CPU per image, host pixels against device pixels
set -x
cd /root/repo
OUT=gpurun_out/r06_feed_gpu.txt
timeout 1500 python -m pytest tests/test_feed_gpu.py -q -x 2>&1 | tail -25 > gpurun_out/r06_feed_gpu_tests.txt
cat gpurun_out/r06_feed_gpu_tests.txt
{
echo "# tools/feeder_rate.py, one feeder, thread backend, bs=64, 640x480 JPEGs -> 416x416 'train' mode with mix-up"
timeout 600 python tools/feeder_rate.py --workers 4,10 --backends thread --native 1 --pixels host,gpu --batches 16
echo "# eight feeder processes at once (the host side of an eight-GPU node on this pod's 16 cores)"
timeout 600 python tools/feeder_rate.py --feeders 8 --workers 10 --backends thread --pixels host --batches 12
timeout 600 python tools/feeder_rate.py --feeders 8 --workers 10 --backends thread --pixels gpu --batches 12
timeout 600 python tools/feeder_rate.py --feeders 8 --workers 4 --backends thread --pixels gpu --batches 12
} > $OUT 2>&1
cat $OUT
