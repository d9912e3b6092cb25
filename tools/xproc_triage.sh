#!/bin/bash
# The cross-process matrix of tools/xproc_triage.hip on ONE GPU: every victim alone, then beside each aggressor process.
#   bash tools/xproc_triage.sh [seconds per victim]   -> gpurun_out/r04/xproc_triage.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-4}
OUT=$R/gpurun_out/r04/xproc_triage.txt
mkdir -p $(dirname $OUT)
VICTIMS="copy memcpy lds ldsdma regs c3h"
{
echo "# victims alone"
for v in $VICTIMS; do $R/tools/xproc_triage victim $v $T; done
for a in c3h mfma copy; do
  echo "# aggressor process: $a"
  $R/tools/xproc_triage aggressor $a $((T * 7 + 6)) > /tmp/aggr.txt 2>&1 &
  AP=$!
  sleep 2
  for v in $VICTIMS; do $R/tools/xproc_triage victim $v $T; done
  wait $AP
  cat /tmp/aggr.txt
done
} 2>&1 | tee $OUT
# engine-level victims beside the stand-alone conv3_h aggressor (12 forwards of 96 patches each, digests must all agree):
#   igemm: every conv on conv_igemm (no LDS-DMA, no counted vmcnt); f32: conv_wino2 + conv_nin (LDS-DMA, counted waits)
{
for tag in igemm f32; do
  if [ $tag = igemm ]; then export DET_OPTS='{"winograd": 0, "nin_gemm": 0}'; else export DET_OPTS='{}'; fi
  echo "# engine victim $tag alone"
  DET_MODES=0 python $R/tools/determinism_check.py 96 12 $tag 2>&1 | cut -c1-150
  echo "# engine victim $tag beside aggressor c3h"
  $R/tools/xproc_triage aggressor c3h 40 > /tmp/aggr.txt 2>&1 &
  AP=$!
  sleep 2
  DET_MODES=0 python $R/tools/determinism_check.py 96 12 $tag 2>&1 | cut -c1-150
  kill $AP 2>/dev/null; wait $AP 2>/dev/null
done
} 2>&1 | tee -a $OUT
