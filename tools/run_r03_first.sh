#!/bin/bash
# first GPU pass of round 3: new parity tests first (fail fast), then the whole GPU suite, then the bench line
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests/test_bench_config_train_gpu.py tests/test_bf16_gpu.py tests/test_rccl_gpu.py tests/test_messi_gpu.py tests/test_decode_gpu.py -m gpu -x -q -s > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/new_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 1500 $O/bench.err
python -m pytest tests -m gpu -q -x --deselect tests/test_bench_config_train_gpu.py --deselect tests/test_rccl_gpu.py > $O/all_tests.log 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/all_tests.log
# probes for the bf16 rewrite: where configs[4]'s time goes per layer, and what LDS-DMA does with out-of-range lanes
python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --csv $O/layers_c5_bf16.csv > $O/layers_c5_bf16.txt 2>&1; tail -4 $O/layers_c5_bf16.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/glds_probe tools/glds_probe.hip && /tmp/glds_probe | tee $O/glds_probe.txt
python tools/layer_profile.py --batch 32 --size 416 --precision f32_wino --csv $O/layers_c2_wino.csv > $O/layers_c2_wino.txt 2>&1; tail -4 $O/layers_c2_wino.txt
