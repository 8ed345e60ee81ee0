#!/bin/bash
# round 3: the four-wave / two-per-CU Winograd kernel against the eight-wave one
O=gpurun_out/r03b; mkdir -p $O
export Y3_WINO_KERNEL=4
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_forward_gpu.py tests/test_bench_config_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -k "wino or forward or bs32 or pipeline" > $O/tests_k4.log 2>&1; echo "tests with Y3_WINO_KERNEL=4 rc=$?" | tee -a $O/summary.txt
tail -4 $O/tests_k4.log
for k in 8 4 8 4; do
  Y3_WINO_KERNEL=$k timeout 300 python tools/layer_profile.py --batch 32 --size 416 --precision f32_wino --iters 20 --csv $O/layers_k${k}_$RANDOM.csv > $O/layers_k$k.txt 2>&1
  echo "kernel $k: $(tail -3 $O/layers_k$k.txt | head -1)" | tee -a $O/summary.txt
done
unset Y3_WINO_KERNEL
for sk in 0 1; do
  Y3_WINO_KERNEL=4 Y3_CONV_WINO_STREAMK=$sk timeout 300 python tools/layer_profile.py --batch 32 --size 416 --precision f32_wino --iters 20 --csv $O/layers_k4_sk$sk.csv > $O/layers_k4_sk$sk.txt 2>&1
  echo "kernel 4 streamk=$sk: $(tail -3 $O/layers_k4_sk$sk.txt | head -1)" | tee -a $O/summary.txt
done
