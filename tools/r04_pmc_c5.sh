R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_c5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- python $R/tools/pmc_layers.py c5 > $O/pmc_sq.log 2>&1
cd $R
python tools/pmc_layers_summary.py $O/pmc_layers_c5.json $O/pmc_sq/p_counter_collection.csv > $O/pmc_layers_c5.txt 2>&1
rm -rf $O/pmc_sq
python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --csv $O/layers_c5.csv > $O/layers_c5.log 2>&1; tail -4 $O/layers_c5.log
