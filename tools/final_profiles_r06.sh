# Round-6 measurement set (run on the GPU box through gpurun; results are copied into profiles/r06_* afterwards).
# Usage: bash tools/final_profiles_r06.sh <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-final_r06}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout 300"
# 1. the PMC passes first (their json must exist before bench.py can report `traffic` for this build)
$T rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $O/pmc_mem -o p -- python $R/tools/pmc_layers.py > $O/pmc_mem.log 2>&1
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- python $R/tools/pmc_layers.py > $O/pmc_sq.log 2>&1
$T rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $O/pmc_mem_c5 -o p -- python $R/tools/pmc_layers.py c5 > $O/pmc_mem_c5.log 2>&1
cd $R
python tools/pmc_layers_summary.py $O/pmc_layers.json $O/pmc_mem/p_counter_collection.csv $O/pmc_sq/p_counter_collection.csv > $O/pmc_layers.txt 2>&1
python tools/pmc_traffic_layers.py $O/pmc_layers.json conv_wino,wino44_input $O/pmc_traffic_wino.json > $O/pmc_traffic_wino.log 2>&1
python tools/pmc_layers_summary.py $O/pmc_layers_c5.json $O/pmc_mem_c5/p_counter_collection.csv > $O/pmc_layers_c5.txt 2>&1
python tools/pmc_traffic_layers.py $O/pmc_layers_c5.json conv_bf16x,conv_bf16p $O/pmc_traffic_bf16.json > $O/pmc_traffic_bf16.log 2>&1
cp $O/pmc_traffic_wino.json profiles/r06_pmc_traffic_wino.json; cp $O/pmc_traffic_bf16.json profiles/r06_pmc_traffic_bf16.json
# 2. the driver's line (default flags), then each workload on its own
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --streams 1 --no-secondary > $O/bench_n1_one_stream.json 2> $O/bench_n1_one_stream.err
$T python tools/streams_ab.py > $O/streams_ab.txt 2>&1
$T python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
$T python bench.py --workload c4 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
$T python bench.py --workload c4 --head-only --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c4_head.json 2> $O/bench_c4_head.err
$T python tools/layer_profile.py --precision f32_wino --csv $O/layers_c2_wino.csv > $O/layers_c2_wino.log 2>&1
$T python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --csv $O/layers_c5_bf16.csv > $O/layers_c5_bf16.log 2>&1
$T python tools/wino44_bench.py > $O/wino44_bench.txt 2>&1
$T python tools/postproc_bench.py > $O/postproc_bench.txt 2>&1
$T python tools/detect_profile.py > $O/detect_profile.txt 2>&1
# 3. kernel statistics of the same commands
cd /tmp
# (--streams 1: whole-batch launches, the ones `roofline.avg_launch_ms` is about; the default two-stream command beside it)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/prof.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2s -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/prof_2s.log 2>&1
python $R/tools/trace_gaps.py $O/prof/p_kernel_trace.csv > $O/trace_gaps.json 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python $R/bench.py --workload c4 --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_c4.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o p -- python $R/bench.py --workload c5 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_c5.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_detect -o p -- python $R/tools/detect_profile.py > $O/prof_detect.log 2>&1
rm -f $O/prof*/p_kernel_trace.csv $O/pmc_*/p_counter_collection.csv $O/pmc_*/*kernel_trace* 2>/dev/null; du -sh $O; cut -c1-300 $O/bench_n1.json; cat $O/trace_gaps.json; cat $O/pmc_traffic_wino.log $O/pmc_traffic_bf16.log
