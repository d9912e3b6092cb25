#!/bin/bash
# Timing-only ablations of fold_border (csrc/conv5_h.hpp: FB_ABL; results wrong by design).
#   build container:  bash tools/fb_abl.sh build      GPU box:  bash tools/fb_abl.sh run
cd "$(dirname "$0")/.."
P=dcscn-super-resolution_amd
MASKS="0 1 2 4 8 15"
if [ "$1" = build ]; then
    mkdir -p tools/abl
    for m in $MASKS; do
        ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -fno-slp-vectorize -DFB_ABL=$m \
              -I include -c $P/csrc/conv5_h.hip -o /tmp/fb_$m.o 2>&1 | grep -E "error"
          hipcc --offload-arch=gfx950 -shared -fPIC $(ls $P/build/*.o | grep -v conv5_h.o) /tmp/fb_$m.o -o tools/abl/libdcscn_fb_$m.so ) &
    done
    wait
else
    cp $P/libdcscn_hip.so /tmp/libdcscn_keep.so
    for m in $MASKS; do
        cp tools/abl/libdcscn_fb_$m.so $P/libdcscn_hip.so
        echo "FB_ABL=$m"
        bash tools/fx_prof.sh L7x4 C5 2>&1 | grep -E "==|fold_border"
    done
    cp /tmp/libdcscn_keep.so $P/libdcscn_hip.so
fi
