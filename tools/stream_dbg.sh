#!/bin/bash
# Timing probe of feat_stream: a library built with -DSTREAM_DBG records, for workgroup 0, the shader clock of every wave at
# four points of 64 steps; tools/stream_dbg.py prints per-role compute / wait times.  build here, run on the GPU box.
R=$(cd $(dirname $0)/.. && pwd)
P=$R/dcscn-super-resolution_amd
mkdir -p $R/tools/abl
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DSTREAM_DBG ${EXTRA} -I $R/include -c $P/csrc/feat_stream.hip -o $R/tools/abl/fs_dbg.o || exit 1
  objs=$(ls $P/build/*.o | grep -v feat_stream.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/tools/abl/fs_dbg.o -o $R/tools/abl/libdcscn_dbg.so || exit 1
else
  mkdir -p $R/gpurun_out
  DCSCN_STREAM_DBG=$R/gpurun_out/stream_dbg.txt DCSCN_LIB=$R/tools/abl/libdcscn_dbg.so python $R/tools/stream_check.py 0 2>&1 | grep "ms"
  python $R/tools/stream_dbg.py $R/gpurun_out/stream_dbg.txt
fi
