"""SURVEY.md section 4 (iv) on ONE GPU: the N-rank run of the bench workload gives, bit for bit, the output of the 1-rank run.

bench.py --strong shards one seeded global batch into contiguous shards (dcscn-super-resolution_amd/shard.py); with
DCSCN_BENCH_SHARE_GPU=1 the ranks share device 0 and rendezvous over gloo, so the multi-rank code path (shard bounds,
per-rank engines, barrier + max-over-ranks timing, rank-ordered gather) runs on a single-GPU box.  Every output patch is
hashed; the digest of the global batch in patch order must not depend on the number of ranks.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_ARGS = ["--strong", "--patches", "96", "--steps", "1", "--warmup", "1", "--check-output", "--no-cpu-baseline", "--no-host-path",
              "--no-layer-by-layer"]


def _run(world, port):
    env = dict(os.environ, DCSCN_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + BENCH_ARGS
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + BENCH_ARGS
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_two_and_three_ranks_reproduce_the_one_rank_output():
    one = _run(1, 0)
    assert one["scaling"] == "strong" and one["n_gpus"] == 1 and one["config"]["global_patches"] == 96
    for world, port in ((2, 29631), (3, 29632)):
        many = _run(world, port)
        assert many["n_gpus"] == world and many["config"]["global_patches"] == 96
        assert many["output_sha256"] == one["output_sha256"], "%d-rank output differs from the 1-rank output" % world


def test_uneven_shards():
    """97 patches over 2 ranks: 49 + 48."""
    global BENCH_ARGS
    saved = list(BENCH_ARGS)
    try:
        BENCH_ARGS[BENCH_ARGS.index("96")] = "97"
        one = _run(1, 0)
        two = _run(2, 29633)
        assert one["output_sha256"] == two["output_sha256"]
        assert two["config"]["patches_per_gpu"] == 49
    finally:
        BENCH_ARGS[:] = saved
