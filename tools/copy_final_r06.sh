#!/bin/bash
# copies a measurement set written by tools/final_profiles_r06.sh (gpurun_out/<dir>) into profiles/r06_*
S=gpurun_out/${1:-final_r06}; P=profiles
cp $S/bench_n1.json $P/r06_bench_n1.json
cp $S/bench_n1_one_stream.json $P/r06_bench_n1_one_stream.json
cp $S/bench_c5.json $P/r06_bench_c5_bf16_608.json
cp $S/bench_c4.json $P/r06_bench_c4_n1.json
cp $S/bench_c4_head.json $P/r06_bench_c4_head_n1.json
cp $S/prof/p_kernel_stats.csv $P/r06_bench_kernel_stats.csv
cp $S/prof_2s/p_kernel_stats.csv $P/r06_bench_kernel_stats_two_streams.csv
cp $S/prof_c5/p_kernel_stats.csv $P/r06_bench_c5_kernel_stats.csv
cp $S/prof_c4/p_kernel_stats.csv $P/r06_train_c4_kernel_stats.csv
cp $S/prof_detect/p_kernel_stats.csv $P/r06_detect_kernel_stats.csv
cp $S/layers_c2_wino.csv $P/r06_layers_bs32_416_wino.csv
cp $S/layers_c5_bf16.csv $P/r06_layers_bs16_608_bf16.csv
cp $S/pmc_layers.txt $P/r06_pmc_layers.txt
cp $S/pmc_layers_c5.txt $P/r06_pmc_layers_c5.txt
cp $S/pmc_traffic_wino.json $P/r06_pmc_traffic_wino.json
cp $S/pmc_traffic_bf16.json $P/r06_pmc_traffic_bf16.json
cp $S/postproc_bench.txt $P/r06_postproc_bench.txt
cp $S/streams_ab.txt $P/r06_streams_ab.txt
cp $S/trace_gaps.json $P/r06_trace_gaps.json
cp $S/wino44_bench.txt $P/r06_wino44_bench.txt
