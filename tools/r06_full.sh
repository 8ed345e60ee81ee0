#!/bin/bash
# the whole GPU suite, then the round's measurement set, in one call
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r06_full_tests.txt
cat gpurun_out/r06_full_tests.txt
bash tools/final_profiles_r06.sh ${1:-final_r06c} 2>&1 | tail -30
