#!/bin/bash
# round 3, step i: the matrix-pipe stem kernel (y3_conv_stem.h) against the thread-per-pixel one, parity tests, train-script test
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_bf16_gpu.py tests/test_forward_gpu.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "conv/bf16/forward tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests_conv.log
for cfg in "Y3_STEM_PK=1" "Y3_STEM_MFMA=1"; do
  env $cfg timeout 300 python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20 --csv $O/layers_c5_${cfg//=/_}.csv > $O/layers_c5_${cfg//=/_}.txt 2>&1
  echo "c5 $cfg: stem $(sed -n 2p $O/layers_c5_${cfg//=/_}.csv) | $(tail -2 $O/layers_c5_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
  env $cfg timeout 300 python tools/layer_profile.py --batch 32 --size 416 --precision f32_wino --iters 20 --csv $O/layers_c2_${cfg//=/_}.csv > $O/layers_c2_${cfg//=/_}.txt 2>&1
  echo "c2 $cfg: stem $(sed -n 2p $O/layers_c2_${cfg//=/_}.csv) | $(tail -2 $O/layers_c2_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_train_script_gpu.py tests/test_train_gpu.py -m gpu -x -q > $O/tests_train.log 2>&1; echo "train tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests_train.log
