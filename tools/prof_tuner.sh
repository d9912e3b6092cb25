#!/bin/bash
# PMC counters for a tuner binary: bash tools/prof_tuner.sh ./tools/wino_tune <tag>
BIN=$1; TAG=${2:-tuner}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $R/$BIN > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $R/$BIN > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_mem -- $R/$BIN > $OUT/mem.log 2>&1
cd $R && python - <<PY
import csv, glob, collections
for d in ("pmc_sq", "pmc_sq2", "pmc_mem"):
    fs = glob.glob("$OUT/%s/**/*_counter_collection.csv" % d, recursive=True)
    if not fs:
        print(d, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "dcscn" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(k)
        for n, v in sorted(c.items()):
            print("    %-28s %14.4g  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
