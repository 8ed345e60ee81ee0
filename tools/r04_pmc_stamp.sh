# the two HBM-traffic PMC passes only (c2 Winograd kernels, c5 bf16 kernels), stamped with the kernel sources' hash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-pmc_stamp}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout 60"
$T rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $O/pmc_mem -o p -- python $R/tools/pmc_layers.py > $O/pmc_mem.log 2>&1
cd $R; python tools/pmc_layers_summary.py $O/pmc_layers.json $O/pmc_mem/p_counter_collection.csv > $O/pmc_layers_mem.txt 2>&1
python tools/pmc_traffic_layers.py $O/pmc_layers.json conv_wino $O/pmc_traffic_wino.json
cd /tmp
$T rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $O/pmc_mem_c5 -o p -- python $R/tools/pmc_layers.py c5 > $O/pmc_mem_c5.log 2>&1
cd $R; python tools/pmc_layers_summary.py $O/pmc_layers_c5.json $O/pmc_mem_c5/p_counter_collection.csv > $O/pmc_layers_c5.txt 2>&1
python tools/pmc_traffic_layers.py $O/pmc_layers_c5.json conv_bf16x,conv_bf16p $O/pmc_traffic_bf16.json
rm -rf $O/pmc_mem $O/pmc_mem_c5
