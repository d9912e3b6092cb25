#!/bin/bash
# Timing-only ablation builds of the streamed kernels: libdcscn with feat_stream.hip compiled at STREAM_ABL = 1..4
# (feat_stream.hpp), linked against the other objects of the shipped build.  Run on the GPU box:
#   bash tools/stream_abl.sh build   (here, cross-compiles)    bash tools/stream_abl.sh run   (on the box)
R=$(cd $(dirname $0)/.. && pwd)
P=$R/dcscn-super-resolution_amd
mkdir -p $R/tools/abl
if [ "$1" = build ]; then
  for n in ${ABLS:-1 2 3 4}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -DSTREAM_ABL=$n -I $R/include -c $P/csrc/feat_stream.hip -o $R/tools/abl/fs$n.o || exit 1
    objs=$(ls $P/build/*.o | grep -v feat_stream.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/tools/abl/fs$n.o -o $R/tools/abl/libdcscn_abl$n.so || exit 1
  done
else
  python $R/tools/stream_check.py 0 2>&1 | grep "ms"
  for n in ${ABLS:-1 2 3 4}; do echo "STREAM_ABL=$n"; DCSCN_LIB=$R/tools/abl/libdcscn_abl$n.so python $R/tools/stream_check.py 0 2>&1 | grep "ms"; done
fi
