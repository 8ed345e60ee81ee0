# Round-2 measurement set (run on the GPU box through gpurun; results are copied into profiles/r02_* by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final5_r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
python $R/tools/trace_gaps.py $O/prof/p_kernel_trace.csv > $O/trace_gaps.json 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv conv_wino_f32_kernel 173350000 $O/pmc_traffic_wino.json 100000 > $O/pmc_traffic_wino.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq/p_counter_collection.csv > $O/pmc_sq_summary.txt 2>&1
cd $R
python tools/layer_profile.py --precision f32_wino --csv $O/layers_wino.csv > $O/layers_wino.log 2>&1
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --workload c4 --steps 4 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload c4 --head-only --steps 4 --warmup 2 > $O/bench_c4_head.json 2> $O/bench_c4_head.err
rm -f $O/prof/p_kernel_trace.csv $O/pmc_*/p_counter_collection.csv $O/pmc_*/*kernel_trace* 2>/dev/null; du -sh $O; cat $O/bench_n1.json | cut -c1-400; cat $O/trace_gaps.json
