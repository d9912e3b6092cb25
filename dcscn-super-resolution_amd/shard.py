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

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()

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
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return Group(rank, world, local_rank, dist)
