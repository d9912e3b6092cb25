"""Multi-GPU path on CPU: the shard arithmetic and a world_size-2 gloo run of the gather used by
evaluate.py (the data path itself has no collective: ranks own disjoint images)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def test_shard_bounds_partition():
    from dcscn_amd import shard
    for n in (0, 1, 5, 8, 1024, 1027):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_bounds(4, 2, 2)


def test_longest_first_assignment_and_split_rule():
    from dcscn_amd import shard
    costs = [900, 100, 400, 400, 250, 50, 700]
    for world in (1, 2, 3, 8):
        plan = shard.assign_longest_first(costs, world)
        assert sorted(i for m in plan for i in m) == list(range(len(costs)))         # a partition
        loads = [sum(costs[i] for i in m) for m in plan]
        assert max(loads) - min(loads) <= max(costs)                                 # LPT bound
        assert plan == shard.assign_longest_first(costs, world)                      # deterministic
    assert shard.assign_longest_first(costs, 2) == [[0, 1, 3], [2, 4, 5, 6]]          # 900+400+100 | 700+400+250+50
    # Set5 on 8 GPUs is split by (image, transform); BSD100 by image; no ensemble -> by image
    assert shard.split_ensemble(5, 8, 8) and shard.split_ensemble(5, 3, 8)
    assert not shard.split_ensemble(100, 8, 8) and not shard.split_ensemble(5, 8, 1) and not shard.split_ensemble(5, 1, 8)


_ENSEMBLE_WORKER = r"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, %r)
from dcscn_amd import shard, imaging
g = shard.init_from_env(backend="gloo")
rng = np.random.default_rng(3)
digests = []
for h, w in ((24, 40), (31, 17)):
    x = rng.uniform(0, 255, (h, w, 1)).astype(np.float32)
    x2 = rng.uniform(0, 255, (2 * h, 2 * w, 1)).astype(np.float32)
    def forward_one(a, b):                      # stand-in for the device forward: any deterministic float32 function
        return (b * np.float32(0.75) + np.repeat(np.repeat(a, 2, 0), 2, 1) * np.float32(0.25) + np.float32(a.shape[0])).astype(np.float32)
    y = g.ensemble_mean(x, x2, 8, forward_one, imaging.flip)
    if g.rank == 0:
        assert y.dtype == np.float64 and y.shape == x2.shape
        digests.append(hashlib.sha256(y.tobytes()).hexdigest())
    else:
        assert y is None                        # the mean lives on rank 0 only
if g.rank == 0:
    print("DIGEST", g.rank, " ".join(digests))
g.close()
"""


def _run_world(tmp_path, script_text, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / ("worker_%d.py" % world)
    script.write_text(script_text)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    return outs


def test_sharded_self_ensemble_equals_the_one_rank_mean_bit_for_bit(tmp_path):
    """The (image, transform) partition of evaluate.py on gloo with 1, 2 and 3 ranks: the float64 ensemble mean rank 0 ends
    up with is bit-identical to the single-process one (transforms gathered there, summed in the reference's order)."""
    ref = None
    for world in (1, 2, 3):
        outs = _run_world(tmp_path, _ENSEMBLE_WORKER % ROOT, world)
        digs = {ln.split(" ", 2)[2] for o in outs for ln in o.splitlines() if ln.startswith("DIGEST")}
        assert len(digs) == 1, outs                            # rank 0 holds the mean
        ref = ref or digs
        assert digs == ref, "world %d differs from world 1" % world


_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
from dcscn_amd import shard
g = shard.init_from_env(backend="gloo")
items = ["img_%%03d" %% i for i in range(7)]
mine = [(name, g.rank) for name in g.my_items(items)]
allr = g.gather(mine)
assert [a[0] for a in allr] == items, allr
assert sorted(set(a[1] for a in allr)) == list(range(g.world))
print("rank", g.rank, "ok", len(mine))
g.close()
"""


def test_gather_world_size_2_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "ok 4" in outs[0] and "ok 3" in outs[1]
