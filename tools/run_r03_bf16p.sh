#!/bin/bash
# round 3: the pipelined bf16 kernel (counted vmcnt, raw barriers): correctness under each forced tile, then the sweep
O=gpurun_out/r03d; mkdir -p $O
for t in A B none; do
  if [ $t = none ]; then unset Y3_BF16X_TILE; else export Y3_BF16X_TILE=$t; fi
  timeout 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "conv_matches or 608" > $O/tests_tile_$t.log 2>&1; echo "bf16 tests tile=$t rc=$?" | tee -a $O/summary.txt
  tail -3 $O/tests_tile_$t.log
done
unset Y3_BF16X_TILE
for cfg in "Y3_BF16X=0" "Y3_BF16X_TILE=A" "Y3_BF16X_TILE=B" "Y3_BF16X_TILE=C" "Y3_BF16X=1"; do
  env $cfg timeout 300 python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20 --csv $O/layers_${cfg//=/_}.csv > $O/layers_${cfg//=/_}.txt 2>&1
  echo "$cfg: $(tail -3 $O/layers_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
done
