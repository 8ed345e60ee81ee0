# round 4, experiment 1: V-plane layout of conv_wino44 (LDS bank conflicts of the fragment reads)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/exp1; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q -k "f4x4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== base lib"; Y3_LIB_PATH=$R/tools/_probe/lib_base.so python tools/wino44_bench.py 2>&1 | tee $O/bench_base.txt
echo "== new lib"; python tools/wino44_bench.py 2>&1 | tee $O/bench_new.txt
echo "== probe old"; Y3_LIB_PATH=$R/tools/_probe/lib_probe_old.so python tools/wino44_probe.py 2>&1 | tee $O/probe_old.txt
echo "== probe new"; Y3_LIB_PATH=$R/tools/_probe/lib_probe.so python tools/wino44_probe.py 2>&1 | tee $O/probe_new.txt
Y3_WINO44=2 python tools/layer_profile.py --precision f32_wino --csv $O/layers_all44.csv > $O/layers_all44.log 2>&1; tail -4 $O/layers_all44.log
python tools/layer_profile.py --precision f32_wino --csv $O/layers_def.csv > $O/layers_def.log 2>&1; tail -4 $O/layers_def.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- python $R/tools/pmc_layers.py > $O/pmc_sq.log 2>&1
cd $R
python tools/pmc_layers_summary.py $O/pmc_layers.json $O/pmc_sq/p_counter_collection.csv > $O/pmc_layers.txt 2>&1
grep -E "wino44" $O/pmc_layers.txt | head -4
rm -rf $O/pmc_sq
