"""Multi-GPU sharding of a batch of independent robot instances.

The OSC path has no cross-instance term (/root/reference/irl_control/osc.py:120-210 touches one
robot), so a node runs one process per GPU, each owning a contiguous slice of the batch; there is
NO data-path collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests) is used only to agree on the elapsed time (max over ranks) and to sum the processed
steps — the "final throughput reduction" of BASELINE.json.
"""
from typing import Tuple


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` instances for `rank`; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_throughput(steps_done: float, elapsed_s: float, device=None):
    """-> (total steps over all ranks, max elapsed over ranks, whole-job steps/s)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(steps_done), float(elapsed_s), float(steps_done) / float(elapsed_s)
    t = torch.tensor([float(steps_done)], dtype=torch.float64, device=device)
    e = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    return float(t[0]), float(e[0]), float(t[0]) / float(e[0])
