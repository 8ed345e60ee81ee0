O=gpurun_out/w8; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_bench_config_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo tests=$?; tail -2 $O/tests.log
for h in ${HS:-0 1}; do
  Y3_WINO8=$h python bench.py --no-cpu-baseline > $O/bench_w$h.json 2>$O/bench_w$h.err; python -c "
import json;d=json.loads(open('$O/bench_w$h.json').read().strip().splitlines()[-1]);print('wino8 $h', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  Y3_WINO8=$h python tools/layer_profile.py --precision f32_wino --csv $O/layers_w$h.csv > $O/layers_w$h.log 2>&1; tail -3 $O/layers_w$h.log
done
python - <<'PY'
import csv
a=list(csv.DictReader(open('gpurun_out/w8/layers_w0.csv'))); b=list(csv.DictReader(open('gpurun_out/w8/layers_w1.csv')))
for x,y in zip(a,b):
    if int(x['layer']) in (3,6,11,13,28,30,45,47,53,61,69):
        print(x['layer'],x['cin'],x['cout'],x['ms'],y['ms'], '%.1f%%'%(100*(float(y['ms'])/float(x['ms'])-1)))
PY
