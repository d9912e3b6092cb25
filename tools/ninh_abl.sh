#!/bin/bash
# Timing-only ablations of conv_nin_h (csrc/conv_nin_h.hpp: NINH_ABL; results wrong by design): what keeps A1 || B1 below the read ceiling.
#   in the build container:  bash tools/ninh_abl.sh build      -> tools/abl/libdcscn_ninh_<mask>.so
#   on the GPU box:          bash tools/ninh_abl.sh run        -> one line per build (bench model and c-DCSCN x2, 1024 patches)
cd "$(dirname "$0")/.."
P=dcscn-super-resolution_amd
MASKS=${MASKS:-"0 1 2 4 7 8"}
if [ "$1" = build ]; then
    mkdir -p tools/abl
    for m in $MASKS; do
        (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -fno-slp-vectorize -DNINH_ABL=$m \
              -I include -c $P/csrc/conv_nin_h.hip -o /tmp/ninh_$m.o 2>&1 | grep -E " error" 
        hipcc --offload-arch=gfx950 -shared -fPIC $(ls $P/build/*.o | grep -v conv_nin_h.o) /tmp/ninh_$m.o -o tools/abl/libdcscn_ninh_$m.so) &
    done
    wait
else
    cp $P/libdcscn_hip.so /tmp/libdcscn_keep.so
    for m in $MASKS; do
        cp tools/abl/libdcscn_ninh_$m.so $P/libdcscn_hip.so
        echo "NINH_ABL=$m"
        for c in "C3" "L7 "; do python tools/bench_configs.py --ops --steps 10 --only "$c" 2>&1 | grep -E "B1\+A1"; done
    done
    cp /tmp/libdcscn_keep.so $P/libdcscn_hip.so
fi
