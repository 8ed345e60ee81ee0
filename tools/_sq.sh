O=gpurun_out/w8sq; mkdir -p $O; R=$PWD; cd /tmp; export TMPDIR=/tmp
for h in ${HS:-0 1}; do
Y3_WINO8=$h timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/sq$h -o p -- python $R/tools/pmc_layers.py > $R/$O/sq$h.log 2>&1
Y3_WINO8=$h timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/$O/g$h -o p -- python $R/tools/pmc_layers.py > $R/$O/g$h.log 2>&1
python $R/tools/pmc_layers_summary.py $R/$O/layers_w$h.json $R/$O/sq$h/p_counter_collection.csv $R/$O/g$h/p_counter_collection.csv > $R/$O/layers_w$h.txt 2>&1
rm -f $R/$O/sq$h/p_counter_collection.csv $R/$O/g$h/p_counter_collection.csv
done
tail -3 $R/$O/g1.log
