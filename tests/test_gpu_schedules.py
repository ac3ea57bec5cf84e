"""Scenario schedules inside the tick kernel (-m gpu): every group of the engine runs TestBackup2B / TestFigure82C
on its own peers in ONE launch; the oracle runs the same op table; every word must agree."""
import numpy as np
import pytest

from raft_amd import _abi
from tests import schedules
from tests.parity import diff_states
from tests.test_schedules import test_ops_one_by_one as _ops_one_by_one, test_schedule_validation as _validation

pytestmark = pytest.mark.gpu


def _mk(**kw):
    from raft_amd import Engine
    return Engine(_abi.make_config(**kw))


def _oracle(**kw):
    from oracle.binding import OracleEngine
    return OracleEngine(_abi.make_config(**kw))


def test_ops_one_by_one_on_the_engine():
    _ops_one_by_one(_mk)
    _validation(_mk)


@pytest.mark.parametrize("name,net", [("Backup2B", {}), ("Figure82C", {}),
                                      ("Figure82C", dict(req_drop_permille=100, reply_drop_permille=100, reply_delay_permille=667))])
def test_schedule_word_for_word(name, net):
    """Small engines, several launches (the schedule's position carries over), all state compared."""
    ops, ticks = schedules.ALL[name]()
    kw = dict(n_groups=200, n_peers=5, log_cap=256, seed=31, apply_ring_cap=512, **net)
    o, e = _oracle(**kw), _mk(**kw)
    o.set_schedule(ops)
    e.set_schedule(ops)
    done = 0
    for n in (137, 1, 400, ticks - 538):
        o.step_mt(n, 8)
        e.step(n)
        done += n
        d = diff_states(o, e)
        assert not d, "after tick %d: %s" % (done, "; ".join(d[:5]))
    assert np.array_equal(o.state_hash(), e.state_hash())
    for g in (0, 57, 199):
        for p in range(5):
            assert o.drain_applied(g, p) == e.drain_applied(g, p)


@pytest.mark.parametrize("name", ["Backup2B", "Figure82C"])
def test_schedule_on_16384_groups_in_one_launch(name):
    ops, ticks = schedules.ALL[name]()
    G = 16384
    kw = dict(n_groups=G, n_peers=5, log_cap=256, seed=32)
    e = _mk(**kw)
    e.set_schedule(ops)
    e.step(ticks)
    h, fl = e.state_hash(), e.read("gflags")
    total = dict.fromkeys(e.counters(), 0)
    for first in range(0, G, 4096):
        o = _oracle(**dict(kw, n_groups=4096, group_base=first))
        o.set_schedule(ops)
        o.step_mt(ticks, 16)
        assert np.array_equal(o.state_hash(), h[first:first + 4096]), "groups %d.." % first
        assert np.array_equal(o.read("gflags"), fl[first:first + 4096])
        for k, v in o.counters().items():
            total[k] += v
        o.close()
    assert total == e.counters()
    if name == "Backup2B":                                   # the forced divergence of test_test.go:387-389, 412-415 in every group
        assert total["ae_conflicts"] >= G
        commit = e.read("commit")
        assert ((commit == 102).all(axis=1)).mean() > 0.85


@pytest.mark.parametrize("name", sorted(schedules.MORE) + ["ReElection2A"])
def test_more_scenarios_on_16384_groups_in_one_launch(name):
    """The other scenario tables (tests/schedules.py): one launch on 16,384 groups, every group's digest and flags
    against the oracle's."""
    if name == "ReElection2A":
        P, (ops, ticks, _) = 3, schedules.reelection_2a()
    else:
        P, fn = schedules.MORE[name]
        ops, ticks, _ = fn()
    G = 16384
    kw = dict(n_groups=G, n_peers=P, log_cap=64, seed=33)
    e = _mk(**kw)
    e.set_schedule(ops)
    e.step(ticks)
    h, fl = e.state_hash(), e.read("gflags")
    total = dict.fromkeys(e.counters(), 0)
    for first in range(0, G, 4096):
        o = _oracle(**dict(kw, n_groups=4096, group_base=first))
        o.set_schedule(ops)
        o.step_mt(ticks, 16)
        assert np.array_equal(o.state_hash(), h[first:first + 4096]), "groups %d.." % first
        assert np.array_equal(o.read("gflags"), fl[first:first + 4096])
        for k, v in o.counters().items():
            total[k] += v
        o.close()
    assert total == e.counters()


@pytest.mark.parametrize("name", ["Figure8Unreliable2C", "UnreliableAgree2C"])
def test_unreliable_scenarios_on_16384_groups(name):
    """The two scenarios that change the net half way (cfg.setlongreordering / cfg.setunreliable between two launches):
    16,384 groups, every digest and flag word against the oracle's."""
    if name == "Figure8Unreliable2C":
        (ops, ticks, at), rates, cap = schedules.figure8_unreliable_2c(), (100, 100, 667), 1024
    else:
        (ops, ticks, at), rates, cap = schedules.unreliable_agree_2c(), (0, 0, 0), 512
    G = 16384
    kw = dict(n_groups=G, n_peers=5, log_cap=cap, seed=34, req_drop_permille=100, reply_drop_permille=100)
    e = _mk(**kw)
    e.set_schedule(ops)
    e.step(at)
    e.set_unreliable(*rates)
    e.step(ticks - at)
    h, fl = e.state_hash(), e.read("gflags")
    total = dict.fromkeys(e.counters(), 0)
    for first in range(0, G, 4096):
        o = _oracle(**dict(kw, n_groups=4096, group_base=first))
        o.set_schedule(ops)
        o.step_mt(at, 16)
        o.set_unreliable(*rates)
        o.step_mt(ticks - at, 16)
        assert np.array_equal(o.state_hash(), h[first:first + 4096]), "groups %d.." % first
        assert np.array_equal(o.read("gflags"), fl[first:first + 4096])
        for k, v in o.counters().items():
            total[k] += v
        o.close()
    assert total == e.counters()
