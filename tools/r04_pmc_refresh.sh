# re-take the PMC traffic passes (stamped with the kernel sources' hash) and the bench lines after a source change
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-pmc_refresh}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout 300"
$T rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $O/pmc_mem -o p -- python $R/tools/pmc_layers.py > $O/pmc_mem.log 2>&1
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- python $R/tools/pmc_layers.py > $O/pmc_sq.log 2>&1
$T rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $O/pmc_mem_c5 -o p -- python $R/tools/pmc_layers.py c5 > $O/pmc_mem_c5.log 2>&1
cd $R
python tools/pmc_layers_summary.py $O/pmc_layers.json $O/pmc_mem/p_counter_collection.csv $O/pmc_sq/p_counter_collection.csv > $O/pmc_layers.txt 2>&1
python tools/pmc_traffic_layers.py $O/pmc_layers.json conv_wino $O/pmc_traffic_wino.json
python tools/pmc_layers_summary.py $O/pmc_layers_c5.json $O/pmc_mem_c5/p_counter_collection.csv > $O/pmc_layers_c5.txt 2>&1
python tools/pmc_traffic_layers.py $O/pmc_layers_c5.json conv_bf16x,conv_bf16p $O/pmc_traffic_bf16.json
cp $O/pmc_traffic_wino.json profiles/r04_pmc_traffic_wino.json; cp $O/pmc_traffic_bf16.json profiles/r04_pmc_traffic_bf16.json
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rm -rf $O/pmc_mem $O/pmc_sq $O/pmc_mem_c5
python -c "import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['roofline']['traffic'], d['roofline']['traffic_source'][:80])"
