#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace + stats, then PMC counters in their own passes
# (never combined with other trace domains).  Run on the GPU box from the repository root:
#   bash tools/rocprof_bench.sh <tag>        -> gpurun_out/prof/<tag>/..., summaries in gpurun_out/prof_summary/
set -x
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-opt-in --no-host-path"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1
cd $R && python tools/summarize_rocprof.py $OUT $R/gpurun_out/prof_summary $TAG
ls -la $R/gpurun_out/prof_summary
