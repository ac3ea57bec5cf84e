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
    from raft_amd.distributed import all_reduce_words, shard_base
    e = OracleEngine(_abi.make_config(n_groups=G, group_base=shard_base(rank, G), **KW))
    e.step(TICKS)
    e.safety_check()
    res = all_reduce_words(e.counters(), e.safety_summary(), torch.device("cpu"))
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
    from raft_amd.distributed import flags_of
    (counters, summary), h0 = torch.load(out)
    h1 = torch.load(out + ".1")
    whole = OracleEngine(_abi.make_config(n_groups=2 * G, **KW))
    whole.step(TICKS)
    whole.safety_check()
    assert counters == whole.counters()
    assert summary == whole.safety_summary() and summary["flagged_groups"] > 0
    assert flags_of(summary) == whole.safety_flags()
    assert h0 + h1 == whole.state_hash().tolist()


def test_summary_is_identity_without_process_group():
    from raft_amd import _abi
    from raft_amd.distributed import all_reduce_words, flags_of, join_comm
    c = {k: i * 3 for i, k in enumerate(_abi.COUNTERS)}
    s = {k: 0 for k in _abi.SUMMARY}
    s.update(groups_election_safety=4, groups_state_machine=2, groups_ref_panic=1, flagged_groups=5)
    assert all_reduce_words(c, s) == (c, s)
    assert flags_of(s) == (0x29, 5)
    assert join_comm(None) is False
