"""Sharding of independent work items (LR patches, images) across the GPUs of one node.

The DCSCN forward pass has no cross-image coupling, so multi-GPU is an embarrassingly parallel
partition: rank r of W takes a contiguous shard, weights are replicated (a few MB), and nothing is
exchanged on the data path.  ``torch.distributed`` (RCCL on GPUs, gloo on CPU) is used only to
gather small per-item results (PSNR values, timings) on rank 0.
"""

import os


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced [begin, end) of ``rank``: the first ``n % world`` ranks get one extra item."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def assign_longest_first(costs, world):
    """Greedy longest-processing-time assignment of items with the given costs to ``world`` ranks: items are taken in
    order of decreasing cost (ties: lower index first) and each goes to the currently least-loaded rank (ties: lower
    rank).  Returns ``[sorted item indices of rank 0, of rank 1, ...]`` -- deterministic, the same on every rank."""
    if world <= 0:
        raise ValueError("bad world %d" % world)
    load = [0] * world
    mine = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += costs[i]
        mine[r].append(i)
    return [sorted(m) for m in mine]


def split_ensemble(n_items, world, self_ensemble):
    """with fewer images than 2 x ranks and a self-ensemble, the work items are (image, transform) pairs
    -- every rank takes transforms t = rank, rank + world, ... of EVERY image -- instead of whole images."""
    return world > 1 and self_ensemble > 1 and n_items < 2 * world


class Group:
    """The process group of this launch (a single process when not launched by torch.distributed.run)."""

    def __init__(self, rank=0, world=1, local_rank=0, dist=None):
        self.rank, self.world, self.local_rank, self._dist = rank, world, local_rank, dist

    def my_items(self, items):
        begin, end = shard_bounds(len(items), self.rank, self.world)
        return items[begin:end]

    def gather(self, local_results):
        """Concatenate every rank's list in rank order (= original item order); valid on every rank."""
        if self.world == 1:
            return list(local_results)
        parts = [None] * self.world
        self._dist.all_gather_object(parts, list(local_results))
        return [r for part in parts for r in part]

    def gather_root(self, local_results):
        """As gather, but only rank 0 receives: the concatenated list there, None on every other rank (full-resolution images: 8
        ranks need not hold 8 copies -- ADVICE r03 / VERDICT r04)."""
        if self.world == 1:
            return list(local_results)
        parts = [None] * self.world if self.rank == 0 else None
        self._dist.gather_object(list(local_results), parts, dst=0)
        return [r for part in parts for r in part] if self.rank == 0 else None

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()

    def ensemble_mean(self, image, bicubic, n, forward_one, flip):
        """Distributed self-ensemble of ONE image (DCSCN.py:559-573): transform t runs on rank t % world, the float32 results
        are gathered ON RANK 0, which forms the float64 mean in the reference's order t = 0 .. n-1 (np.zeros float64, +=, / n);
        every other rank gets None (it has nothing to do with the image any more: metrics and files are rank 0's).
        ``forward_one(x[h, w, 1], x2[sh, sw, 1]) -> [sh, sw, 1] float32``; ``flip(image, t, invert)`` = util.flip."""
        import numpy as np

        def mine():
            parts = []
            for t in range(self.rank, n, self.world):
                y = forward_one(np.ascontiguousarray(flip(image, t)), np.ascontiguousarray(flip(bicubic, t)))
                parts.append((t, np.ascontiguousarray(flip(np.asarray(y, np.float32), t, invert=True))))
            return parts
        parts = mine()
        gathered = self.gather_root(parts)
        if gathered is None:
            return None
        every = dict(gathered)
        out = np.zeros(every[0].shape, dtype=np.float64)
        for t in range(n):
            out += every[t]
        return out / n

    def close(self):
        if self.world > 1 and self._dist.is_initialized():
            self._dist.barrier()
            self._dist.destroy_process_group()


def init_from_env(backend=None):
    """Join the group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return Group()
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        # DCSCN_SHARE_GPU=1 (test rig: all ranks on device 0) cannot use RCCL: one communicator rank per device
        backend = "nccl" if torch.cuda.is_available() and os.environ.get("DCSCN_SHARE_GPU") != "1" else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        if os.environ.get("DCSCN_SHARE_GPU") == "1":
            local_rank = 0
        dist.init_process_group(backend, rank=rank, world_size=world)
    return Group(rank, world, local_rank, dist)
