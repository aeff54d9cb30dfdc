"""Sharding of a batch of independent frames over one process per GPU.

The reference scales by running more Kafka consumers in one group (kafka/consumer.go:23,
docker-compose.yaml:86-96): messages are independent, nothing is exchanged.  The same holds here:
frames are independent units, so there is NO data-path collective.  torch.distributed is used only
for (a) a barrier and the max-over-ranks of the timed region and (b) an atomic counter in the
rendezvous store that implements pull scheduling ("work stealing") for mixed-size batches.
"""
import os
import time


def round_robin(n_items, rank, world):
    """Static partition for uniform batches: image_index mod nGPU."""
    return list(range(rank, n_items, world))


def lpt_order(costs):
    """Largest-first order (longest processing time first) for pull scheduling of mixed sizes."""
    return sorted(range(len(costs)), key=lambda i: (-costs[i], i))


def frame_cost(w, h):
    """Bytes moved for one frame by the full pipeline: 4WH read + 4WH watermark + outputs."""
    return 8 * w * h + 4 * 1024 * 768 + 4 * 200 * 200


class WorkQueue:
    """Pull scheduling over a shared counter.  Every rank walks the same largest-first order and
    claims the next unclaimed chunk with an atomic add in the torch.distributed store, so a rank that
    finishes early simply claims more: work stealing falls out of pull scheduling.  With no process
    group (single process) the counter is local."""

    def __init__(self, n_items, chunk=1, key="ipx_queue", store=None):
        self.n, self.chunk, self.key, self.store = n_items, max(1, chunk), key, store
        self._local = 0

    def claim(self):
        """-> range of item positions (in the shared order), or None when the queue is empty."""
        if self.store is not None:
            hi = self.store.add(self.key, self.chunk)   # atomic fetch-add across ranks
            lo = hi - self.chunk
        else:
            lo = self._local
            self._local += self.chunk
        if lo >= self.n:
            return None
        return range(lo, min(self.n, lo + self.chunk))


def default_store():
    """The store of the default process group (TCPStore under torchrun), or None."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.distributed_c10d._get_default_store()
    except Exception:
        pass
    return None


def aggregate(units, seconds):
    """Whole-job numbers: units summed over ranks, seconds = max over ranks."""
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            u = torch.tensor([float(units)], dtype=torch.float64)
            t = torch.tensor([float(seconds)], dtype=torch.float64)
            dist.all_reduce(u, op=dist.ReduceOp.SUM)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(u.item()), float(t.item())
    except ImportError:
        pass
    return float(units), float(seconds)


def init_from_env():
    """One process per GPU under torch.distributed.run; gloo carries barrier / reduce / store only."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, local_rank, world
