R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/layer_profile.py --precision f32_wino --csv $O/layers_wino.csv > $O/layers_wino.log 2>&1
python tools/layer_profile.py --csv $O/layers_f32.csv > $O/layers_f32.log 2>&1
for p in f32 f32_wino f32_bf16x6; do python tools/train_bench.py --batch 64 --steps 3 --precision $p > $O/train_$p.log 2>&1; done
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
find $O -name "*.csv" | head -20; rm -f $O/prof/*/*kernel_trace.csv $O/pmc_*/*/*kernel_trace* 2>/dev/null; du -sh $O
