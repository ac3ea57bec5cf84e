"""World-size-2 run of the sharded path over gloo on CPU: each rank advances its
own block of groups (here with the CPU checker standing in for a GPU's engine),
the summary all-reduce is the product's, and the result must equal one
unsharded run."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(n_peers=5, log_cap=128, seed=2, req_drop_permille=100, reply_drop_permille=100,
          start_period=5, start_max=100, part_period=100, part_jitter=20)
G, TICKS = 48, 1500


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.binding import OracleEngine
    from raft_amd import _abi
    from raft_amd.distributed import all_reduce_summary, shard_base
    e = OracleEngine(_abi.make_config(n_groups=G, group_base=shard_base(rank, G), **KW))
    e.step(TICKS)
    e.safety_check()
    fl, nfl = e.safety_flags()
    res = all_reduce_summary(e.counters(), fl, nfl, torch.device("cpu"))
    if rank == 0:
        torch.save((res, e.state_hash().tolist()), out)
    else:
        torch.save(e.state_hash().tolist(), out + ".1")
    dist.destroy_process_group()


def test_two_ranks_equal_one_unsharded_run(tmp_path, oracle_lib):
    from oracle.binding import OracleEngine
    from raft_amd import _abi
    out = str(tmp_path / "r0.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (counters, flags, nflagged), h0 = torch.load(out)
    h1 = torch.load(out + ".1")
    whole = OracleEngine(_abi.make_config(n_groups=2 * G, **KW))
    whole.step(TICKS)
    whole.safety_check()
    assert counters == whole.counters()
    assert (flags, nflagged) == whole.safety_flags()
    assert h0 + h1 == whole.state_hash().tolist()


def test_summary_is_identity_without_process_group():
    from raft_amd import _abi
    from raft_amd.distributed import all_reduce_summary
    c = {k: i * 3 for i, k in enumerate(_abi.COUNTERS)}
    assert all_reduce_summary(c, 0x29, 5, torch.device("cpu")) == (c, 0x29, 5)
