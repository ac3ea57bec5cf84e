"""The named benchmark configurations, HIP engine vs CPU oracle (-m gpu).

Reduced group counts run the whole schedule on both sides, in two launches, and compare per-group digests
of every Raft-visible word, counters and flags after each.  The full-size runs compare every group after
one launch: all seeded streams key on the global group id, so the oracle replays the groups slice by slice
(on the host cores, a few seconds per configuration)."""
import os

import numpy as np
import pytest

from raft_amd import _abi

pytestmark = pytest.mark.gpu

CFG2 = dict(n_peers=5, log_cap=1024, seed=1, start_period=5, start_max=1000)
CFG3 = dict(n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5,
            start_max=1000, part_period=100, part_jitter=20)
# crash / restart through the snapshots every 70 ticks (test_test.go:893-919), every leader takes the Starts
# (test_test.go:697-703) so that deposed leaders pile up entries that are truncated later (test_test.go:387-389)
CFG4 = dict(n_peers=7, log_cap=4096, seed=4, req_drop_permille=50, reply_drop_permille=50, start_period=10, start_max=3500,
            start_mode=_abi.START_ALL, churn_period=70, churn_disconnect_permille=200, churn_restart_permille=500,
            churn_crash_permille=200)
CFG5 = dict(n_peers=3, log_cap=256, seed=5, req_drop_permille=10, reply_drop_permille=10, start_period=5, start_max=200)


def _oracle(**kw):
    from oracle.binding import OracleEngine
    return OracleEngine(_abi.make_config(**kw))


def _engine(**kw):
    from raft_amd import Engine
    return Engine(_abi.make_config(**kw))


def _same(o, e):
    assert np.array_equal(o.state_hash(), e.state_hash())
    assert o.counters() == e.counters()
    assert np.array_equal(o.read("gflags"), e.read("gflags"))
    assert o.safety_check() == e.safety_check()


@pytest.mark.parametrize("layout", [_abi.LOG_ROWS, _abi.LOG_COLUMNS])
@pytest.mark.parametrize("name,kw,groups,ticks", [
    ("cfg2", CFG2, 4096, 10000),
    ("cfg3", CFG3, 3000, 20000),
    ("cfg4", CFG4, 300, 40000),
    ("cfg5", CFG5, 8192, 4000),
])
def test_whole_schedule_reduced(name, kw, groups, ticks, layout):
    """Both log layouts (rows: entry i of all peers together; columns: per-peer arrays) give the oracle's words."""
    o, e = _oracle(n_groups=groups, **kw), _engine(n_groups=groups, log_layout=layout, **kw)
    half = ticks // 2
    for n in (half, ticks - half):
        o.step_mt(n, 8)
        e.step(n)
        _same(o, e)
    c = e.counters()
    assert c["group_ticks"] == groups * ticks
    assert c["entries_committed"] > 0
    if name == "cfg2":
        assert c["entries_committed"] == groups * 1000 and e.safety_flags() == (0, 0)
    if name == "cfg4":
        assert c["ae_conflicts"] > 0 and c["persist_entries"] > 0


THREADS = min(16, os.cpu_count() or 1)


def _every_group(kw, groups, ticks, width, engine_base=0):
    """One engine launch over all groups; the oracle replays them `width` groups at a time."""
    e = _engine(n_groups=groups, group_base=engine_base, **kw)
    e.step(ticks)
    h = e.state_hash()
    fl = e.read("gflags")
    total = dict.fromkeys(e.counters(), 0)
    for first in range(0, groups, width):
        n = min(width, groups - first)
        o = _oracle(n_groups=n, group_base=engine_base + first, **kw)
        o.step_mt(ticks, THREADS)
        assert np.array_equal(o.state_hash(), h[first:first + n]), "groups %d..%d" % (first, first + n)
        assert np.array_equal(o.read("gflags"), fl[first:first + n]), "flags of groups %d..%d" % (first, first + n)
        for k, v in o.counters().items():
            total[k] += v
        o.close()
    assert total == e.counters()
    return e


def test_cfg3_full_size_every_group():
    """65,536 five-peer groups, 10 % + 10 % loss, rolling partitions, 20,000 ticks in one launch."""
    e = _every_group(CFG3, 65536, 20000, 8192)
    c = e.counters()
    assert c["group_ticks"] == 65536 * 20000
    # properties that need no oracle, on the groups in which the reference's code stays inside its own invariants
    # (no flag raised): every live peer has applied all it knows committed, commitIndex is inside the log, all
    # peers that applied the same number of entries hold the same apply digest
    assert c["entries_committed"] <= c["entries_started"]
    ok = e.read("gflags") == 0
    commit, applied, loglen = e.read("commit")[ok], e.read("applied")[ok], e.read("loglen")[ok]
    assert ok.sum() > 30000 and (applied == commit).all() and (commit < loglen).all()
    digest = e.read("applyhash")[ok]
    same = applied == applied[:, :1]
    assert (digest[same] == np.broadcast_to(digest[:, :1], digest.shape)[same]).all()


def test_cfg5_shard_of_a_million_groups():
    """One rank's 131,072 three-peer groups of the 1,048,576-group job (rank 5 of 8): the result of a
    shard depends only on its global group ids."""
    _every_group(CFG5, 131072, 4000, 16384, engine_base=5 * 131072)


def test_target_shard_131072x5_every_group():
    """One rank's share of the target job -- 1,048,576 five-peer groups on 8 GPUs = 131,072 x 5 per GPU -- under the
    benchmark workload (10 % + 10 % loss, rolling partitions, 1000 entries, 20,000 ticks), as rank 5 of 8."""
    e = _every_group(CFG3, 131072, 20000, 8192, engine_base=5 * 131072)
    c = e.counters()
    assert c["group_ticks"] == 131072 * 20000
    assert c["entries_committed"] > 0.7 * 131072 * 1000      # partitions and loss keep some groups from finishing


@pytest.mark.parametrize("layout", [_abi.LOG_ROWS, _abi.LOG_AUTO])
def test_cfg4_full_size_every_group(layout):
    """16,384 seven-peer groups, 4096-entry logs, crash / restart churn; with the layout the engine picks for such a
    configuration (columns) and with rows."""
    e = _every_group(dict(CFG4, log_layout=layout), 16384, 40000, 2048)
    assert e.counters()["ae_conflicts"] > 0


def test_target_shard_in_columns():
    """The benchmark workload at full size in the layout it does not pick by itself."""
    _every_group(dict(CFG3, log_layout=_abi.LOG_COLUMNS), 65536, 20000, 8192)
