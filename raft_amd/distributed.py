"""Cross-GPU reduction of the engine's counters and safety summary.

Groups never exchange messages (reference: src/raft/config.go:70 builds a
private network per group), so the data path needs no collective.  The only
exchange is one all-reduce (sum) of RG_N_COUNTERS + RG_N_SUMMARY 64-bit words,
latency bound: the counters, the number of groups per safety-flag bit (a bit is
set somewhere in the job iff its count is non-zero), the flagged groups and
their share of the work.

The product path is `rg_allreduce_summary` of the C ABI (include/raftgpu.h):
one ncclAllReduce over xGMI inside the library, on the engine's stream, with a
communicator made by `rg_comm_init`.  `join_comm` below carries the
ncclUniqueId from rank 0 to the others over an existing torch.distributed
group (any backend: it is 128 bytes of host data).  `all_reduce_words` is the
same reduction over torch.distributed itself, for ranks that cannot form an
RCCL communicator: CPU tests over gloo, and several ranks rehearsing on one GPU.
"""
import torch
import torch.distributed as dist

from ._abi import COMM_ID_BYTES, COUNTERS, FLAG_NAMES, N_COUNTERS, SUMMARY


def shard_base(rank, groups_per_rank):
    """Global id of a rank's first group: contiguous blocks, rank-major."""
    return rank * groups_per_rank


def flags_of(summary):
    """(OR of the flag words, flagged groups) as rg_safety_flags reports them, from a (reduced) summary."""
    bits = sum(1 << i for i, n in enumerate(FLAG_NAMES) if summary["groups_" + n])
    return bits, summary["flagged_groups"]


def _distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_reduce_words(counters, summary, device=None):
    """(counters, summary) summed over all ranks of the default process group; identity when not distributed."""
    if not _distributed():
        return dict(counters), dict(summary)
    vals = [counters[k] for k in COUNTERS] + [summary[k] for k in SUMMARY]
    t = torch.tensor(vals, dtype=torch.int64).to(device or torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    v = t.tolist()
    return dict(zip(COUNTERS, v[:N_COUNTERS])), dict(zip(SUMMARY, v[N_COUNTERS:]))


def join_comm(engine, device=None):
    """Makes the engine's RCCL communicator over the ranks of the default process group (rg_comm_unique_id on rank 0,
    the id broadcast, rg_comm_init everywhere) and returns True -- on EVERY rank, or on none: if any rank fails, all
    drop theirs and return False, so that no rank is left waiting in a collective the others never enter."""
    if not _distributed():
        return False
    rank, world = dist.get_rank(), dist.get_world_size()
    device = device or torch.device("cpu")
    ok, uid = 1, bytes(COMM_ID_BYTES)
    if rank == 0:
        try:
            uid = type(engine).comm_unique_id()
        except Exception:
            ok = 0
    t = torch.tensor(list(uid) + [ok], dtype=torch.int64).to(device)
    dist.broadcast(t, 0)
    v = t.tolist()
    ok, uid = v[-1], bytes(v[:-1])
    if ok:
        try:
            engine.comm_init(rank, world, uid)
        except Exception:
            ok = 0
    agreed = torch.tensor([ok], dtype=torch.int64).to(device)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    if int(agreed.item()) == 0:
        try:
            engine.comm_destroy()
        except Exception:
            pass
        return False
    return True
