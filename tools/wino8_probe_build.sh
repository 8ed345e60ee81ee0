#!/bin/bash
# knock-out builds of the eight-wave Winograd kernel -> tools/_probe/libyolo355_ko<N>.so (not committed)
set -e
cd $(dirname $0)/..; C=yolov3_tensorflow_amd/csrc; mkdir -p tools/_probe
OBJS=$(ls $C/*.o | grep -v y3_conv_wino.o)
for k in ${KOS:-0 1 2 3 4 8 7 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DY3_WINO8_KO=$k ${EXTRA} -c $C/y3_conv_wino.hip -o tools/_probe/wino_ko$k.o &
done
wait
for k in ${KOS:-0 1 2 3 4 8 7 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_probe/libyolo355_ko$k.so $OBJS tools/_probe/wino_ko$k.o
done
ls tools/_probe/*.so
