#!/bin/bash
# instruction-cache counters of the narrow nets' streamed kernels (feat3_stream / feat_stream): tools/rocprof_l7.sh <tag>
TAG=${1:-l7}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/bench_configs.py --steps 2 --only L7"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_ic -- $B > $OUT/pmc_ic.log 2>&1
cd $R && python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_ic/**/*_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "stream" not in k: continue
    acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()): print("   %-24s %.4g" % (c, x))
PY
