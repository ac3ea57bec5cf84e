"""Relay launches of the tick kernel (-m gpu): more sets of groups than resident blocks, the launch cut into
chunks that blocks take from a queue, each chunk continuing from the state another block stored.  The engine is
told the chip is tiny (RAFTGPU_SLOT_WAVES) so that small cases take that path; whatever the cut, every word has to
match the tick-by-tick oracle."""
import pytest

from raft_amd import _abi
from tests.parity import diff_states

pytestmark = pytest.mark.gpu

LOSSY = dict(n_peers=5, log_cap=300, seed=11, req_drop_permille=100, reply_drop_permille=100, start_period=3, start_max=280,
             part_period=60, part_jitter=20)
CHURN = dict(n_peers=7, log_cap=300, seed=12, req_drop_permille=50, reply_drop_permille=50, start_period=4, start_max=290,
             start_mode=_abi.START_ALL, churn_period=35, churn_disconnect_permille=200, churn_restart_permille=500,
             churn_crash_permille=200, account_bytes=1)
THREE = dict(n_peers=3, log_cap=96, seed=13, req_drop_permille=10, reply_drop_permille=10, start_period=5, start_max=90)


@pytest.mark.parametrize("kw,groups,slot_waves,chunks,launches,sets", [
    (LOSSY, 200, 8, 3, (700, 1, 299, 500), 5),      # 5 sets of 48 groups, room for 2 blocks
    (LOSSY, 500, 16, 7, (1500,), 11),               # 11 sets, room for 4 blocks, 7 chunks
    (CHURN, 150, 4, 4, (400, 400, 401), 5),         # 5 sets of 36 groups, room for 1 block
    (THREE, 1000, 12, 5, (250, 1000), 13),          # three peers: 80 groups a set (84 would split cache lines)
    (LOSSY, 96, 8, 4, (600,), 0),                   # 2 sets, room for 2 blocks: plain launch
    (LOSSY, 200, 8, 300, (600, 7), 5),              # a boundary (and a new dealing of the groups) every 2 ticks, then every tick
    (CHURN, 150, 4, 64, (640,), 5),                 # ... with crashes and restarts, every 10 ticks
])
def test_relay_matches_oracle(monkeypatch, capfd, kw, groups, slot_waves, chunks, launches, sets):
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    monkeypatch.setenv("RAFTGPU_SLOT_WAVES", str(slot_waves))
    monkeypatch.setenv("RAFTGPU_RELAY_CHUNKS", str(chunks))
    monkeypatch.setenv("RAFTGPU_TRACE_LAUNCH", "1")
    monkeypatch.setenv("RAFTGPU_DUE_MAX_TICKS", "0")       # the short launches too go through the relay here, not through due launches
    cfg = dict(kw, n_groups=groups)
    o, e = OracleEngine(_abi.make_config(**cfg)), Engine(_abi.make_config(**cfg))
    for n in launches:
        o.step_mt(n, 8)
        e.step(n)
        d = diff_states(o, e)
        assert not d, "; ".join(d[:5])
    assert e.counters() == o.counters()
    assert e.counters()["entries_committed"] > 0
    trace = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("raftgpu: k_tick")]
    assert len(trace) == len(launches) and all("relay_sets=%d " % sets in ln for ln in trace), trace


@pytest.mark.parametrize("kw,groups,slot_waves,chunks,ticks", [
    (CHURN, 150, 4, 4, 1200),       # relay launch: a peer's ring is written by several blocks of one launch
    (CHURN, 40, 0, 0, 1500),        # plain launch
    (LOSSY, 200, 8, 3, 900),
])
def test_apply_streams_match_oracle(monkeypatch, kw, groups, slot_waves, chunks, ticks):
    """applyCh of every peer after ONE long launch with crashes, restarts (replays from index 1) and truncations:
    the engine's rings hold the oracle's streams message for message."""
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    if slot_waves:
        monkeypatch.setenv("RAFTGPU_SLOT_WAVES", str(slot_waves))
        monkeypatch.setenv("RAFTGPU_RELAY_CHUNKS", str(chunks))
    cfg = dict(kw, n_groups=groups, apply_ring_cap=2048)
    o, e = OracleEngine(_abi.make_config(**cfg)), Engine(_abi.make_config(**cfg))
    replays = 0
    for n in (ticks, 37):
        o.step_mt(n, 8)
        e.step(n)
        assert not diff_states(o, e)
        for g in range(groups):
            for p in range(cfg["n_peers"]):
                mo = o.drain_applied(g, p)
                assert mo == e.drain_applied(g, p), (g, p)
                idx = [i for i, _ in mo]
                replays += sum(1 for a, b in zip(idx, idx[1:]) if b == 1)
    if "churn_period" in kw:
        assert replays > 0
