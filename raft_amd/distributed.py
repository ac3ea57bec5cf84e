"""Cross-GPU reduction of the engine's counters and safety flags.

Groups never exchange messages (reference: src/raft/config.go:70 builds a
private network per group), so the data path needs no collective.  The only
exchange is this one all-reduce (sum) of a few hundred bytes, latency bound: the
counters, one word per flag bit (the number of ranks that saw it; non-zero = set,
which is the OR) and the number of flagged groups.  With backend "nccl" this is
RCCL over xGMI; tests run it over gloo on CPU.
"""
import torch
import torch.distributed as dist

from ._abi import COUNTERS, N_COUNTERS

N_FLAG_BITS = 7


def shard_base(rank, groups_per_rank):
    """Global id of a rank's first group: contiguous blocks, rank-major."""
    return rank * groups_per_rank


def pack(counters, flags, flagged_groups, device):
    vals = [counters[k] for k in COUNTERS] + [(flags >> b) & 1 for b in range(N_FLAG_BITS)] + [flagged_groups]
    return torch.tensor(vals, dtype=torch.int64).to(device)


def unpack(t):
    v = t.tolist()
    counters = dict(zip(COUNTERS, v[:N_COUNTERS]))
    flags = sum((1 << i) for i, b in enumerate(v[N_COUNTERS:N_COUNTERS + N_FLAG_BITS]) if b)
    return counters, flags, int(v[N_COUNTERS + N_FLAG_BITS])


def all_reduce_summary(counters, flags, flagged_groups, device):
    """Returns (counters, flags, flagged_groups) over all ranks; identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return dict(counters), flags, flagged_groups
    t = pack(counters, flags, flagged_groups, device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return unpack(t)
