#!/bin/bash
# PMC counters of the split16 harness (tools/h16_tune.hip): bash tools/rocprof_h16.sh <tag> <harness args...>
#   -> gpurun_out/prof_h16/<tag>_pmc.txt  (per-kernel averages: MFMA busy, LDS bank conflicts, wait states, clock)
TAG=${1:-t}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_h16/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- ${H16_BIN:-$R/tools/h16_tune} "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_b -- ${H16_BIN:-$R/tools/h16_tune} "$@" > $OUT/pmc_b.log 2>&1
cd $R && python - "$OUT" "$R/gpurun_out/prof_h16/${TAG}_pmc.txt" <<'PY'
import collections, csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dcscn::" not in k: continue
            k = k.replace("void dcscn::", "").replace("(dcscn::ConvArgs)", "")
            # separate launches of the same kernel with different grids (layers)
            k = "%s grid %s" % (k, r.get("Grid_Size", "?"))
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
            if (d, r["Dispatch_Id"]) not in seen:
                seen.add((d, r["Dispatch_Id"])); dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open(dst, "w") as o:
    for k in sorted(acc):
        c = {name: acc[k][name] / n[k][name] for name in acc[k]}
        t = sum(dur[k]) / len(dur[k])
        line = "%s: avg %.3f ms" % (k, t * 1e-6)
        if "GRBM_GUI_ACTIVE" in c: line += "  clock %.2f GHz" % (c["GRBM_GUI_ACTIVE"] / t)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c: line += "  mfma_busy/(1024 SIMD x GUI/8?) raw %.3g" % c["SQ_VALU_MFMA_BUSY_CYCLES"]
        o.write(line + "\n    " + "  ".join("%s=%.4g" % kv for kv in sorted(c.items())) + "\n")
print(open(dst).read())
PY
