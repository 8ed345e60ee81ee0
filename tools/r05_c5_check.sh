# Round 5: configs[4] after a change - parity (product + forced tiles), per-layer table of the default dispatch, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_c5_check; mkdir -p $O; cd $R
timeout 1500 python -m pytest -x -q -m gpu tests/test_bf16_gpu.py tests/test_conv_variants_gpu.py -k "bf16" 2>&1 | tail -5
python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20 --csv $O/default.csv 2>&1 | grep -E "^total|^k="
python tools/layer_profile.py --batch 8 --size 608 --precision bf16 --iters 20 --csv $O/default_bs8.csv 2>&1 | grep -E "^total|^k="
python bench.py --workload c5 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_c5.json
python - <<PY
import json; d=json.load(open("$O/bench_c5.json")); print("c5:", d["value"], d["ms_per_step"], d["config"].get("streams_calibration_ms"), d["roofline"]["frac"], d["roofline"]["whole_forward_frac"])
PY
