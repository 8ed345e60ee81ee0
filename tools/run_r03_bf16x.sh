#!/bin/bash
# round 3: the LDS-DMA bf16 kernel (tests, tile sweep) + SQ counters of the two Winograd kernels
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -s > $O/tests_bf16.log 2>&1; echo "bf16 tests rc=$?" | tee -a $O/summary.txt
grep -E "bs=16|passed|failed|Error|error" $O/tests_bf16.log | tail -12
for cfg in "Y3_BF16X=0" "Y3_BF16X=1" "Y3_BF16X_TILE=A" "Y3_BF16X_TILE=B" "Y3_BF16X_TILE=C"; do
  env $cfg timeout 300 python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20 --csv $O/layers_${cfg//=/_}.csv > $O/layers_${cfg//=/_}.txt 2>&1
  echo "$cfg: $(tail -3 $O/layers_${cfg//=/_}.txt | tr '\n' ' ')" | tee -a $O/summary.txt
done
timeout 300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?" | tee -a $O/summary.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for k in 8 4; do
  Y3_WINO_KERNEL=$k timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_sq_k$k -o p -- python $R/tools/pmc_layers.py > $R/$O/pmc_sq_k$k.log 2>&1
  echo "pmc k$k rc=$?" | tee -a $R/$O/summary.txt
done
cd $R
ls $O/pmc_sq_k8 $O/pmc_sq_k4 2>/dev/null | head
