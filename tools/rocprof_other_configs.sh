#!/bin/bash
# rocprofv3 evidence for bench.py's other_configs legs (BASELINE C2 = L8 x2 / 256 patches, C5 = L7 x4 DS / 1024 patches): kernel trace +
# stats, then the PMC counters in their own passes (never combined with other trace domains).  On the GPU box, from the repository root:
#   bash tools/rocprof_other_configs.sh <tag>      -> gpurun_out/prof_summary/<tag>_c2_* and <tag>_c5_*
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in C2 C5; do
    c=$(echo $C | tr A-Z a-z)
    OUT=$R/gpurun_out/prof/${TAG}_$c
    mkdir -p $OUT
    B="python $R/tools/bench_configs.py --only $C --steps 3"          # 3 warm-up + 3 timed forwards
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1
    (cd $R && DCSCN_PROF_FORWARDS=6 python tools/summarize_rocprof.py $OUT $R/gpurun_out/prof_summary ${TAG}_$c)
done
ls -la $R/gpurun_out/prof_summary
