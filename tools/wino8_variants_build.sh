#!/bin/bash
# named experiment builds: tools/_probe/libyolo355_<name>.so from "name:flags" pairs
set -e
cd $(dirname $0)/..; C=yolov3_tensorflow_amd/csrc; mkdir -p tools/_probe
OBJS=$(ls $C/*.o | grep -v y3_conv_wino.o)
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f -c $C/y3_conv_wino.hip -o tools/_probe/wino_$n.o &
done
wait
for v in "$@"; do
  n=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_probe/libyolo355_$n.so $OBJS tools/_probe/wino_$n.o
done
