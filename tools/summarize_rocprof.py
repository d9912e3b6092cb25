#!/usr/bin/env python
"""Condenses the rocprofv3 CSV output of tools/rocprof_bench.sh into the small summaries committed under
profiles/ (kernel-trace stats of the dcscn kernels, and per-kernel PMC sums)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    return name.replace("void dcscn::", "").replace("(dcscn::ConvArgs)", "").replace("(dcscn::Cin1Args, int)", "") \
               .replace("(dcscn::DwArgs, long long)", "")


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        rows = [r for r in csv.DictReader(open(stats[0])) if "dcscn::" in r["Name"]]
        with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % ("python bench.py --steps 3 --warmup 1 --no-cpu-baseline" if "DCSCN_PROF_FORWARDS" not in os.environ
                                                                  else "python tools/bench_configs.py --only <config> --steps 3"))
            f.write("# (dcscn kernels only; %s forwards x launches per forward)\n" % os.environ.get("DCSCN_PROF_FORWARDS", "4"))
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                            r["MaxNs"], r["StdDev"]])
    pmc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    dur = collections.defaultdict(float)
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        seen = set()
        for r in csv.DictReader(open(files[0])):
            if "dcscn::" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (os.path.basename(d), r["Dispatch_Id"])
            if os.path.basename(d) == "pmc_sq" and key not in seen:
                seen.add(key)
                calls[k] += 1
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if pmc:
        out = {}
        for k in pmc:
            n = max(calls[k], 1)
            # forwards per profiled run: bench.py --steps 3 --warmup 1 = 4; tools/rocprof_other_configs.sh sets 6 (3 warm-up + 3 timed)
            out[k] = {"dispatches": calls[k], "forwards": int(os.environ.get("DCSCN_PROF_FORWARDS", "4")), "avg_duration_ns_profiled": dur[k] / n}
            out[k].update({c: v / n for c, v in pmc[k].items()})
        with open(os.path.join(dst, "%s_pmc_per_dispatch.json" % tag), "w") as f:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench                                      # csrc_digest: bench.py compares it with the build it runs on
            json.dump({"note": "per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KiB as reported (uncalibrated, "
                               "see MI355X_MICROARCH.md HBM section); SQ_* summed over the chip",
                       "csrc_sha256": bench.csrc_digest(),
                       "kernels": out}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
