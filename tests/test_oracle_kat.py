"""Known answers for the CPU checker, hand-derived from the reference source
(src/raft/raft.go; no golden vectors ship with the reference).  The same
cases run against the HIP engine in tests/test_gpu_api.py."""
import numpy as np
import pytest

from raft_amd import _abi
from tests import kat


@pytest.fixture
def mk(oracle_lib):
    from oracle.binding import OracleEngine
    return lambda **kw: OracleEngine(_abi.make_config(**kw))


@pytest.mark.parametrize("case", sorted(kat.CASES))
def test_known_answer(mk, case):
    kat.CASES[case](mk)


def test_delta_persist_equals_full_persist(mk):
    """persist() copying only the new part of the committed prefix leaves the
    same snapshot arrays as re-encoding all of it (raft.go:116-122)."""
    kw = dict(n_groups=24, n_peers=5, log_cap=256, seed=5, req_drop_permille=80, reply_drop_permille=80,
              start_period=3, start_max=200, start_mode=_abi.START_ALL, churn_period=70,
              churn_disconnect_permille=200, churn_restart_permille=500, churn_crash_permille=200)
    a, b = mk(**kw), mk(**kw)
    b.set_full_persist(True)
    for _ in range(6):
        a.step(500)
        b.step(500)
        assert np.array_equal(a.state_hash(), b.state_hash())
    assert a.counters() == b.counters()


def test_multithreaded_step_is_the_same(mk):
    kw = dict(n_groups=64, n_peers=5, log_cap=128, seed=9, req_drop_permille=100, reply_drop_permille=100,
              start_period=5, start_max=100, part_period=100, part_jitter=20)
    a, b = mk(**kw), mk(**kw)
    a.step(1500)
    b.step_mt(1500, 4)
    assert np.array_equal(a.state_hash(), b.state_hash())
    assert a.counters() == b.counters()


def test_results_do_not_depend_on_sharding(mk):
    """Streams key on the global group id: groups 32..63 of one engine equal
    groups 0..31 of an engine whose group_base is 32."""
    kw = dict(n_peers=5, log_cap=128, seed=4, req_drop_permille=100, reply_drop_permille=100,
              start_period=5, start_max=100, part_period=100, part_jitter=20)
    whole = mk(n_groups=64, **kw)
    shard = mk(n_groups=32, group_base=32, **kw)
    whole.step(1200)
    shard.step(1200)
    assert np.array_equal(whole.state_hash()[32:], shard.state_hash())
    for f in ("term", "commit", "loglen", "applyhash"):
        assert np.array_equal(whole.read(f)[32:], shard.read(f))


CHURN = dict(n_groups=12, n_peers=5, log_cap=256, seed=5, req_drop_permille=50, reply_drop_permille=50, start_period=3,
             start_max=200, start_mode=_abi.START_ALL, churn_period=70, churn_disconnect_permille=200,
             churn_restart_permille=500, churn_crash_permille=200, apply_ring_cap=4096)


def test_apply_stream_of_a_churn_launch_holds_the_replays(mk):
    """Peers restarted by the device-side churn inside ONE launch replay their restored log from index 1
    (raft.go:717, 677-685): the drained stream shows it, and every message is the tester's record of its index
    unless the group was flagged."""
    e = mk(**CHURN)
    e.step(3000)
    replays = 0
    rec, flags = e.read("applyrec"), e.read("gflags")
    for g in range(12):
        for p in range(5):
            msgs = e.drain_applied(g, p)
            idx = [i for i, _ in msgs]
            replays += sum(1 for a, b in zip(idx, idx[1:]) if b == 1)
            assert all(b == a + 1 or b == 1 for a, b in zip(idx, idx[1:]))     # in order within an incarnation
            if flags[g] == 0:
                assert all(rec[g, i] == c for i, c in msgs)
    assert replays > 5
    assert sum(e.counters()[k] for k in ("entries_applied",)) > 0
