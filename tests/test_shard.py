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
