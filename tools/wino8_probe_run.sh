echo "tag            13x512->1024 26x256->512 52x128->256 104x64->128 208x32->64  (us per launch)"
Y3_TAG=wino4 Y3_WINO8=0 python tools/wino8_probe.py 2>&1 | grep -v amdgpu.ids
for k in $VARS; do
  Y3_TAG=w8_$k Y3_LIB_PATH=$PWD/tools/_probe/libyolo355_$k.so python tools/wino8_probe.py 2>&1 | grep -v amdgpu.ids
done
[ -f tools/_probe/libyolo355_clk.so ] && Y3_LIB_PATH=$PWD/tools/_probe/libyolo355_clk.so python tools/wino8_clock_probe.py 2>&1 | grep -v amdgpu.ids
