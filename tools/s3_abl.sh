#!/bin/bash
# Timing-only ablations of feat3_stream (csrc/feat3_stream.hpp: S3_ABL; results wrong by design): which part of a row step costs what.
#   in the build container:  bash tools/s3_abl.sh build      -> tools/abl/libdcscn_s3_<mask>.so
#   on the GPU box:          bash tools/s3_abl.sh run        -> one line per build (c-DCSCN x2, 1024 patches)
cd "$(dirname "$0")/.."
P=dcscn-super-resolution_amd
MASKS="0 31 32 64 63 95 159 255"      # (0 1 2 3 4 8 16 31) + "dbg": the shipped kernel with per-wave shader-clock sums (S3_DBG), printed once
if [ "$1" = build ]; then
    mkdir -p tools/abl
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -fno-slp-vectorize -DS3_DBG \
          -I include -c $P/csrc/feat3_stream.hip -o /tmp/s3_dbg.o 2>&1 | grep -E "error"
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls $P/build/*.o | grep -v feat3_stream.o) /tmp/s3_dbg.o -o tools/abl/libdcscn_s3_dbg.so
    for m in $MASKS; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -fno-slp-vectorize -DS3_ABL=$m \
              -I include -c $P/csrc/feat3_stream.hip -o /tmp/s3_$m.o 2>&1 | grep -E "error" 
        objs=$(ls $P/build/*.o | grep -v feat3_stream.o)
        hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/s3_$m.o -o tools/abl/libdcscn_s3_$m.so
    done
else
    cp $P/libdcscn_hip.so /tmp/libdcscn_keep.so
    for m in $MASKS; do
        cp tools/abl/libdcscn_s3_$m.so $P/libdcscn_hip.so
        echo -n "S3_ABL=$m  "
        python tools/bench_configs.py --ops --only "L7 " 2>&1 | grep feat3
    done
    cp tools/abl/libdcscn_s3_dbg.so $P/libdcscn_hip.so
    DCSCN_S3_DBG=1 python tools/bench_configs.py --steps 1 --only "L7 " 2>&1 | grep S3_DBG | tail -8
    cp /tmp/libdcscn_keep.so $P/libdcscn_hip.so
fi
