O=gpurun_out/w8tcp; mkdir -p $O; R=$PWD; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" "TA_BUFFER_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/$O/p$i -o p -- python $R/tools/pmc_layers.py > $R/$O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python tools/pmc_layers_summary.py $O/layers.json $O/p*/p_counter_collection.csv > $O/layers.txt 2>&1
rm -f $O/p*/p_counter_collection.csv
