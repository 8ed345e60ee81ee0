#!/bin/bash
# per-layer memory-side counters of the forward: byte-weighted DRAM read/write requests (gfx950 events 112/115),
# the request counters FETCH_SIZE is built from, L2 hit/miss.  Separate passes (TCC has 4 counter slots).
O=${1:-gpurun_out/pmc_layers}; R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/$O/p$i -o p -- python $R/tools/pmc_layers.py > $R/$O/p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
cd $R
python tools/pmc_layers_summary.py $O/layers.json $O/p*/p_counter_collection.csv > $O/layers.txt 2>&1
tail -90 $O/layers.txt
