#!/bin/bash
# Shard-size and pass-size sweep of the bench workload on ONE MI355X: the per-GPU rate at the shard sizes a
# strong-scaling run of BASELINE configs[2] would give each rank (1024 / N patches), and the engine's pass size (sub_batch_pixels) at
# the full shard.  Run on the GPU box from the repository root:  bash tools/shard_sweep.sh > gpurun_out/shard_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-opt-in --no-host-path --no-other-configs"
pick='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln); k = d["kernel_ms_per_step"]
        print("%8.2f LR Mpix/s  %8.3f ms/step  3x3 %.3f  nin %.3f  cin1 %.3f  tail %.3f" % (d["value"], d["ms_per_step"],
              d["roofline"]["kernel_ms_per_step"], k.get("conv_nin_h", 0), k.get("conv_cin1", 0), k.get("conv5_h", 0)))'
echo "# shard size (patches per GPU), default pass size (one pass)"
for p in 1024 512 256 128; do printf "patches %5d   " $p; $B --patches $p 2>/dev/null | python -c "$pick"; done
echo "# pass size (sub_batch_pixels, in 48x48 patches) at 1024 patches per GPU"
for s in 32 64 128 256 512; do printf "pass %5d     " $s; $B --patches 1024 --sub-batch-pixels $((s * 2304)) 2>/dev/null | python -c "$pick"; done
