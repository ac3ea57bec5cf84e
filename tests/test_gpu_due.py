"""Due launches (-m gpu): launches of a few ticks run only the groups whose next event falls inside the launch and
leave the others alone, their lazy work (wake-up chains, client Starts) waiting for their next span.  Whatever the
mix of due launches, full launches and host verbs, and however long nothing is read in between, every word has to
match the tick-by-tick oracle once it is read."""
import random

import numpy as np
import pytest

from raft_amd import _abi
from tests.parity import diff_states

pytestmark = pytest.mark.gpu

LOSSY = dict(n_peers=5, log_cap=256, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=200,
             part_period=100, part_jitter=20)
CHURN = dict(n_peers=7, log_cap=512, seed=4, req_drop_permille=50, reply_drop_permille=50, start_period=3, start_max=400,
             start_mode=_abi.START_ALL, churn_period=35, churn_disconnect_permille=200, churn_restart_permille=500,
             churn_crash_permille=200, account_bytes=1)
THREE = dict(n_peers=3, log_cap=96, seed=13, req_drop_permille=10, reply_drop_permille=10, start_period=5, start_max=90, apply_ring_cap=64)
HOSTED = dict(n_peers=3, log_cap=128, seed=21)        # no device client: Starts come from the host


def _pair(**kw):
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    return OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))


def _trace(capfd):
    return [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("raftgpu: k_tick")]


@pytest.mark.parametrize("kw,groups,window,split", [(LOSSY, 700, 0, 0), (LOSSY, 700, 64, 1), (CHURN, 300, 7, 0), (THREE, 2000, 0, 0), (THREE, 150, 64, 0),
                                                    (LOSSY, 6000, 256, 8), (THREE, 9000, 200, 3)])
def test_runs_of_due_launches_between_reads(monkeypatch, capfd, kw, groups, window, split):
    """Stretches of short launches with nothing read in between (so the skipped groups stay behind for the whole
    stretch), cut by long launches that find lagging groups; large windows: more due groups than one round of a block
    holds; split: several blocks share a window's rounds and must agree on its due groups whichever of them runs first."""
    monkeypatch.setenv("RAFTGPU_TRACE_LAUNCH", "1")
    if window:
        monkeypatch.setenv("RAFTGPU_DUE_WIN", str(window))
    if split:
        monkeypatch.setenv("RAFTGPU_DUE_SPLIT", str(split))
    o, e = _pair(n_groups=groups, **kw)
    rng = random.Random(groups)
    t = 0
    for stretch in ([1] * 40, [2, 3, 1, 5, 8, 1, 12, 4] * 6, [300], [1] * 25 + [700] + [7] * 9 + [10, 13, 11], [1] * 120):
        for n in stretch:
            e.step(n)
            t += n
        o.step_mt(sum(stretch), 8)
        d = diff_states(o, e)
        assert not d, "after tick %d: %s" % (t, "; ".join(d[:6]))
        assert np.array_equal(o.state_hash(), e.state_hash())
    tr = _trace(capfd)
    assert sum("k_tick_due" in ln for ln in tr) >= 200 and any("k_tick_due" in ln and "lazy=1" in ln for ln in tr), tr[:5]
    assert any("k_tick ticks=0 " in ln for ln in tr), "no settling launch before a read"
    assert o.safety_check() == e.safety_check()
    assert o.counters()["entries_committed"] > 0
    if kw.get("apply_ring_cap"):
        from raft_amd.engine import RaftGpuError
        for g in rng.sample(range(groups), 20):
            for p in range(kw["n_peers"]):
                res = []
                for x in (o, e):
                    try:
                        res.append(("ok", x.drain_applied(g, p)))
                    except RaftGpuError as err:         # more messages than the ring holds: the newest ones, and the status
                        res.append((err.status, err.partial))
                assert res[0] == res[1]


def test_host_verbs_between_due_launches(monkeypatch, capfd):
    """The tester's way of driving an instance: verbs from the host, one tick at a time.  Without a device client the
    verbs need no settling launch: what lags in skipped groups are wake-up times the verbs neither read nor keep."""
    monkeypatch.setenv("RAFTGPU_TRACE_LAUNCH", "1")
    G, P = 400, 3
    o, e = _pair(n_groups=G, **HOSTED)
    rng = random.Random(5)
    for x in (o, e):
        x.step(120)                                     # elections
    assert not diff_states(o, e)
    capfd.readouterr()
    for rnd in range(60):
        for _ in range(rng.randrange(0, 12)):
            g, p = rng.randrange(G), rng.randrange(P)
            verb = rng.choice(["start", "start", "start", "disconnect", "connect", "crash", "restart", "kill", "get"])
            for x in (o, e):
                if verb == "start":
                    r = x.start(g, p, 1000 * rnd + g)
                elif verb == "disconnect":
                    r = x.set_connected(g, p, False)
                elif verb == "connect":
                    r = x.set_connected(g, p, True)
                elif verb == "crash":
                    r = x.crash(g, p)
                elif verb == "restart":
                    x.crash(g, p)
                    r = x.restart(g, p)
                    x.set_connected(g, p, True)
                elif verb == "kill":
                    r = x.kill(g, p)
                else:
                    r = x.get_state(g, p)
                if x is o:
                    want = r
                else:
                    assert r == want, (verb, g, p)
        n = rng.choice([1, 1, 1, 2, 8, 12])
        o.step(n)
        e.step(n)
    tr = _trace(capfd)
    assert len(tr) == 60 and all("k_tick_due" in ln for ln in tr), tr[:4]      # no full launch, no settling launch
    d = diff_states(o, e)
    assert not d, "; ".join(d[:6])
    assert o.counters()["entries_committed"] > 0


def test_parked_groups_cost_nothing(monkeypatch, capfd):
    """An engine whose groups are all crashed but a few (slots of a tester that wait for a config): their wake words say
    'never', so a one-tick launch looks at 4 bytes per group and runs the live ones."""
    G, P = 3000, 3
    o, e = _pair(n_groups=G, **HOSTED)
    live = [5, 1234, 2999]
    for x in (o, e):
        for g in range(G):
            if g not in live:
                for p in range(P):
                    x.crash(g, p)
        x.step(1)
    for _ in range(300):
        e.step(1)
    o.step(300)
    d = diff_states(o, e)
    assert not d, "; ".join(d[:6])
    lead = o.read("state")[live]
    assert (lead == 2).any(axis=1).all()                # the live groups elected leaders meanwhile


def test_step_time_of_sampled_due_launches(monkeypatch):
    """Of the due launches one in 16 is timed and stands for the others: the launch count stays exact, and the time
    reported is close to what timing every launch reports (same workload, two engines)."""
    from raft_amd import Engine
    kw = dict(n_groups=20000, **LOSSY)
    monkeypatch.setenv("RAFTGPU_DUE_TIME_EVERY", "1")
    every = Engine(_abi.make_config(**kw))
    monkeypatch.delenv("RAFTGPU_DUE_TIME_EVERY")
    sampled = Engine(_abi.make_config(**kw))
    res = []
    for e in (every, sampled):
        e.step(500)
        ms0, n0 = e.step_time_ms()
        for _ in range(403):
            e.step(1)
        ms1, n1 = e.step_time_ms()
        assert n1 - n0 == 403
        res.append((ms1 - ms0) / 403)
    assert np.array_equal(every.state_hash(), sampled.state_hash())
    assert 0.5 * res[0] < res[1] < 2.0 * res[0], res       # (without the weights it would be a sixteenth)
