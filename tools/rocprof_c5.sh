#!/bin/bash
# PMC passes for the small nets (C5 / c-DCSCN L7): gpurun_out/prof/<tag>_c5/...
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/${TAG}_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/bench_configs.py --only C5 --steps 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_inst -- $B > $OUT/pmc_inst.log 2>&1
cd $R && python tools/summarize_rocprof.py $OUT $R/gpurun_out/prof_summary ${TAG}_c5
