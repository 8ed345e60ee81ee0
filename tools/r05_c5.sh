# Round 5, configs[4] (608x608 bs=16 bf16): parity of the new bf16 kernels, then per-layer times with every tile shape forced
# in turn - ALL inside one gpurun call (boxes differ by 3-5 %).   usage: tools/r05_c5.sh [quick]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_c5; mkdir -p $O; cd $R
EXP=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so
LP="python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20"
echo "== parity ==" 
timeout 1200 python -m pytest -x -q -m gpu tests/test_bf16_gpu.py 2>&1 | tail -6
echo "== default dispatch (product library) =="
$LP --csv $O/default.csv 2>&1 | grep -E "^total|^k="
echo "== bs=8 (one of two streams) default =="
python tools/layer_profile.py --batch 8 --size 608 --precision bf16 --iters 20 --csv $O/default_bs8.csv 2>&1 | grep -E "^total|^k="
for t in A B C D E; do
  echo "== 3x3 tile $t =="; Y3_LIB_PATH=$EXP Y3_BF16X_TILE=$t $LP --csv $O/x_$t.csv 2>&1 | grep -E "^total|^k="
done
for t in a b c d e f g; do
  echo "== 1x1 tile $t =="; Y3_LIB_PATH=$EXP Y3_BF16R_TILE=$t $LP --csv $O/r_$t.csv 2>&1 | grep -E "^total|^k="
done
echo "== 1x1 register-staged kernel (round 4) =="; Y3_LIB_PATH=$EXP Y3_BF16R=0 $LP --csv $O/r_old.csv 2>&1 | grep -E "^total|^k="
if [ "$1" != "quick" ]; then
  for t in A B C D E; do Y3_LIB_PATH=$EXP Y3_BF16X_TILE=$t python tools/layer_profile.py --batch 8 --size 608 --precision bf16 --iters 20 --csv $O/x8_$t.csv 2>&1 | grep -E "^total" | sed "s/^/bs8 3x3 $t: /"; done
  for t in a b c d e f g; do Y3_LIB_PATH=$EXP Y3_BF16R_TILE=$t python tools/layer_profile.py --batch 8 --size 608 --precision bf16 --iters 20 --csv $O/r8_$t.csv 2>&1 | grep -E "^total" | sed "s/^/bs8 1x1 $t: /"; done
fi
echo "== bench c5 =="
python bench.py --workload c5 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_c5.json; python - <<PY
import json; d=json.load(open("$O/bench_c5.json")); print(d["value"], d["ms_per_step"], d["config"].get("streams_calibration_ms"), d["roofline"]["frac"])
PY
python tools/r05_tile_table.py $O
