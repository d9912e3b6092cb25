#!/bin/bash
# kernel times of the folded tail's two launches: bash tools/fx_prof.sh [configs ...]   (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/fx
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- L7x4 C5
for c in "$@"; do
    d=$(echo "$c" | tr -c 'A-Za-z0-9\n' _)
    rm -rf "$R/gpurun_out/fx/$d"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/fx/$d" -- python $R/tools/bench_configs.py --only "$c" --steps 20 > "$R/gpurun_out/fx/$d.log" 2>&1 < /dev/null
    f=$(find "$R/gpurun_out/fx/$d" -name "*kernel_stats.csv" | head -1)
    echo "== $c  $(grep ms/step "$R/gpurun_out/fx/$d.log" | cut -c1-75)"
    [ -n "$f" ] && grep -E "conv5_h|fold_border|feat_stream<true|feat3_stream|tail_stream<true" "$f" | cut -d, -f1-4
done
