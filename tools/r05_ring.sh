# Round 5: the 1x1 ring kernel's variants (drain after the epilogue or not, ring depth) against the product, per tile shape,
# inside ONE call.  usage: tools/r05_ring.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_ring; mkdir -p $O; cd $R
LP="python tools/layer_profile.py --batch 16 --size 608 --precision bf16 --iters 20"
EXP=$R/yolov3_tensorflow_amd/csrc/libyolo355_exp.so
echo "== parity with the no-drain builds =="
for v in r_nodrain r_deep_nodrain r_deep; do
  Y3_LIB_PATH=$R/tools/_probe/lib_$v.so timeout 600 python -m pytest -x -q -m gpu tests/test_bf16_gpu.py -k conv_matches 2>&1 | tail -1 | sed "s/^/$v: /"
done
echo "== default (product) =="; $LP --csv $O/default.csv 2>&1 | grep -E "^total|^k="
for v in exp r_nodrain r_deep r_deep_nodrain; do
  L=$R/tools/_probe/lib_$v.so; [ $v = exp ] && L=$EXP
  for t in b c d e f g; do
    Y3_LIB_PATH=$L Y3_BF16R_TILE=$t $LP --csv $O/${v}_$t.csv 2>&1 | grep -E "^k=1" | sed "s/^/$v tile $t: /"
  done
done
python - <<PY
import csv,os
O="$O"
rows=lambda p:[r for r in csv.DictReader(open(p))]
base=rows(O+"/default.csv")
names=[(v,t) for v in ("exp","r_nodrain","r_deep","r_deep_nodrain") for t in "bcdefg"]
data={k:rows("%s/%s_%s.csv"%(O,k[0],k[1])) for k in names if os.path.exists("%s/%s_%s.csv"%(O,k[0],k[1]))}
seen=set()
print("layer cin cout | default | " + " | ".join(v for v in ("exp","nodrain","deep","deep+nodrain")) + "   (each: tiles b c d e f g, us)")
for i,r in enumerate(base):
    if r["k"]!="1": continue
    key=(r["cin"],r["cout"],i in (58,59,60,66,67,68,74))
    if (r["cin"],r["cout"]) in seen: continue
    seen.add((r["cin"],r["cout"]))
    cells=[]
    for v in ("exp","r_nodrain","r_deep","r_deep_nodrain"):
        cells.append(" ".join("%5.1f"%(float(data[(v,t)][i]["ms"])*1e3) if (v,t) in data else "    -" for t in "bcdefg"))
    print("%3d %4s %4s | %5.1f | %s"%(i,r["cin"],r["cout"],float(r["ms"])*1e3," | ".join(cells)))
PY
