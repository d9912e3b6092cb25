set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -- $B > $R/gpurun_out/prof/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- $B > $R/gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_write -- $B > $R/gpurun_out/prof/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_sq -- $B > $R/gpurun_out/prof/pmc_sq.log 2>&1
find $R/gpurun_out/prof -type f | head -50
du -sh $R/gpurun_out/prof
cd $R && python bench.py --steps 10 --warmup 3 | tee gpurun_out/bench2.json
