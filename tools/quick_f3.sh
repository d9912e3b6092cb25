#!/bin/bash
# rebuild only feat3_stream.o and relink (iteration aid; build.py rebuilds everything when any header changes)
cd /root/repo/dcscn-super-resolution_amd
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -I ../include -c csrc/feat3_stream.hip -o build/feat3_stream.o 2>&1 | grep -E "error|feat3_stream.hpp:[0-9]+:1: remark:     (VGPRs:|Scratch|VGPRs Spill)"
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o libdcscn_hip.so && touch build/*.o && touch libdcscn_hip.so && echo relinked
