"""C-ABI verbs of the HIP engine against the CPU checker on random states (-m gpu):
the batched RequestVote / AppendEntries handler kernels, Start batches, the apply
stream, Persister blobs, bulk state access, and state digests at scale."""
import numpy as np
import pytest

from raft_amd import _abi
from tests.parity import diff_states

pytestmark = pytest.mark.gpu


def _pair(**kw):
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    return OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))


def _random_logs(rng, G, P, L, max_term=6):
    """Logs with non-decreasing terms, sentinel at 0, random lengths and commit points."""
    ln = rng.integers(1, L - 8, size=(G, P))
    steps = rng.random((G, P, L)) < 0.25
    terms = np.minimum(np.cumsum(steps, axis=2), max_term).astype(np.int32)
    terms[:, :, 0] = 0
    terms[:, :, 1:] = np.maximum(terms[:, :, 1:], 1)
    cmds = rng.integers(1, 2 ** 62, size=(G, P, L)).astype(np.uint64)
    cmds[:, :, 0] = 0
    commit = (rng.random((G, P)) * ln).astype(np.int32)
    return ln.astype(np.int32), terms, cmds, commit


def _seed_state(engines, rng, G, P, L):
    ln, terms, cmds, commit = _random_logs(rng, G, P, L)
    term = (terms[np.arange(G)[:, None], np.arange(P)[None, :], ln - 1] + rng.integers(0, 3, size=(G, P))).astype(np.int32)
    vote = rng.integers(-1, P, size=(G, P)).astype(np.int32)
    state = rng.integers(0, 3, size=(G, P)).astype(np.int32)
    dead = (rng.random((G, P)) < 0.1).astype(np.int32) * rng.integers(1, 3, size=(G, P)).astype(np.int32)   # crashed or killed
    for e in engines:
        e.write("logterm", terms)
        e.write("logcmd", cmds)
        e.write("loglen", ln)
        e.write("commit", commit)
        e.write("term", term)
        e.write("vote", vote)
        e.write("state", state)
        e.write("dead", dead)
    return ln, terms, commit, term


@pytest.mark.parametrize("peers,seed", [(3, 1), (5, 2), (7, 3)])
def test_append_entries_handler_random_states(peers, seed):
    G, L = 96, 64
    rng = np.random.default_rng(seed)
    o, e = _pair(n_groups=G, n_peers=peers, log_cap=L, seed=seed)
    ln, terms, commit, term = _seed_state((o, e), rng, G, peers, L)
    groups, prs, reqs = [], [], []
    for g in range(G):
        for _ in range(3):                       # several requests per group: handled in array order
            p = int(rng.integers(0, peers))
            kind = rng.random()
            prev = int(rng.integers(0, ln[g, p] + 3)) if kind < 0.8 else int(rng.integers(0, L - 9))
            prev = min(prev, L - 9)
            pt = int(terms[g, p, min(prev, L - 1)]) if rng.random() < 0.6 else int(rng.integers(0, 7))
            t = int(term[g, p] + rng.integers(-1, 3))
            k = int(rng.integers(0, 8))
            ents = [(int(rng.integers(1, 8)), int(rng.integers(1, 2 ** 62))) for _ in range(k)]
            groups.append(g)
            prs.append(p)
            reqs.append((t, int(rng.integers(0, peers)), prev, pt, ents, int(rng.integers(0, L))))
    ro = o.append_entries_batch(groups, prs, reqs)
    re_ = e.append_entries_batch(groups, prs, reqs)
    assert ro == re_
    assert any(r[1] for r in ro) and any(r[2] > 0 for r in ro) and any(r[2] == -1 for r in ro)
    assert not diff_states(o, e)
    assert np.array_equal(o.state_hash(), e.state_hash())


@pytest.mark.parametrize("peers,seed", [(3, 4), (5, 5), (8, 6)])
def test_request_vote_handler_random_states(peers, seed):
    G, L = 128, 48
    rng = np.random.default_rng(seed)
    o, e = _pair(n_groups=G, n_peers=peers, log_cap=L, seed=seed)
    ln, terms, commit, term = _seed_state((o, e), rng, G, peers, L)
    groups, prs, args = [], [], []
    for g in range(G):
        for _ in range(4):
            p = int(rng.integers(0, peers))
            groups.append(g)
            prs.append(p)
            args.append((int(term[g, p] + rng.integers(-1, 3)), int(rng.integers(0, peers)),
                         int(rng.integers(0, L)), int(rng.integers(0, 8))))
    ro = o.request_vote_batch(groups, prs, args)
    re_ = e.request_vote_batch(groups, prs, args)
    assert ro == re_
    assert any(g for _, g in ro) and any(not g for _, g in ro)
    assert not diff_states(o, e)


def test_start_batch_drain_and_snapshots():
    kw = dict(n_groups=40, n_peers=5, log_cap=64, seed=8, apply_ring_cap=64)
    o, e = _pair(**kw)
    o.step(40)
    e.step(40)
    rng = np.random.default_rng(8)
    for rnd in range(6):
        groups = [int(g) for g in rng.integers(0, 40, size=60)]       # repeats: same-group Starts keep their order
        peers = [int(p) for p in rng.integers(0, 2, size=60)]          # peer 0 leads, peer 1 refuses
        cmds = [int(c) for c in rng.integers(1, 2 ** 62, size=60)]
        want = o.start_batch(groups, peers, cmds)
        assert want == e.start_batch(groups, peers, cmds)
        if rnd % 2:                                                    # the array form of the same call
            more = [int(c) + 1 for c in cmds]
            want = o.start_batch(groups, peers, more)
            idx, term, lead = e.start_batch_arrays(np.array(groups), np.array(peers), np.array(more, dtype=np.uint64))
            assert want == [(int(i), int(t), bool(l)) for i, t, l in zip(idx, term, lead)]
        # every group once: the one-pass form of the call (16 bytes per Start each way), leaders and refusing peers mixed
        once = [int(g) for g in rng.permutation(40)]
        pe = [int(p) for p in rng.integers(0, 2, size=40)]
        cm = [int(c) for c in rng.integers(1, 2 ** 62, size=40)]
        want = o.start_batch(once, pe, cm)
        assert want == e.start_batch(once, pe, cm)
        assert any(l for _, _, l in want) and any(not l for _, _, l in want)
        o.step(15)
        e.step(15)
        assert not diff_states(o, e)
        for g in (0, 7, 39):
            for p in (0, 3):
                assert o.drain_applied(g, p) == e.drain_applied(g, p)
    for g, p in ((0, 0), (5, 2), (39, 4)):
        assert o.snapshot_export(g, p) == e.snapshot_export(g, p)
    # move a blob to another peer, restart it from there (Persister swap, config.go:110-126,159-163)
    blob = e.snapshot_export(3, 0)
    for x in (o, e):
        x.snapshot_import(3, 4, blob)
        x.restart(3, 4)
        x.set_connected(3, 4, True)
        x.crash(9, 1)
        x.set_side(11, 2, 1)
    o.step(120)
    e.step(120)
    assert not diff_states(o, e)
    assert e.drain_applied(3, 4) == o.drain_applied(3, 4)
    assert e.safety_check() == o.safety_check()


def test_read_write_roundtrip_every_field():
    from raft_amd import Engine
    e = Engine(_abi.make_config(n_groups=9, n_peers=5, log_cap=32, seed=3))
    rng = np.random.default_rng(0)
    for name, (_, kind) in _abi.FIELDS.items():
        cur = e.read(name)
        if name in ("state",):
            new = rng.integers(0, 3, size=cur.shape)
        elif name == "dead":
            new = rng.integers(0, 3, size=cur.shape)
        elif name in ("stop", "restart", "first", "monalive", "snaphas", "connected", "side"):
            new = rng.integers(0, 2, size=cur.shape)
        elif name == "cnt":
            new = rng.integers(0, 5, size=cur.shape)
        elif cur.dtype == np.uint64:
            new = rng.integers(0, 2 ** 63, size=cur.shape).astype(np.uint64)
        else:
            new = rng.integers(-5, 1000, size=cur.shape)
        e.write(name, new.astype(cur.dtype))
        assert np.array_equal(e.read(name), new.astype(cur.dtype)), name


def test_bad_arguments_are_status_codes():
    from raft_amd import Engine, RaftGpuError
    e = Engine(_abi.make_config(n_groups=2, n_peers=3, log_cap=16))
    for fn in (lambda: e.start(2, 0, 1), lambda: e.start(0, 3, 1), lambda: e.get_state(-1, 0),
               lambda: e.crash(0, 9), lambda: e.append_entries(0, 0, 1, 0, 20, 0, [], 0),
               lambda: e.step(-1)):
        with pytest.raises(RaftGpuError) as ei:
            fn()
        assert ei.value.status in (_abi.ERR_RANGE, _abi.ERR_ARG)
    with pytest.raises(RaftGpuError):
        Engine(_abi.make_config(n_groups=0, n_peers=3, log_cap=16))
    with pytest.raises(RaftGpuError):
        Engine(_abi.make_config(n_groups=1, n_peers=9, log_cap=16))


def test_log_capacity_is_flagged_not_overrun():
    o, e = _pair(n_groups=1, n_peers=3, log_cap=8, seed=1)
    for x in (o, e):
        x.step(30)
        res = [x.start(0, 0, 100 + i) for i in range(10)]
        assert [r[0] for r in res] == [1, 2, 3, 4, 5, 6, 7, -1, -1, -1]
        x.step(30)
        assert x.safety_flags()[0] & _abi.FLAG_LOG_OVERFLOW
    assert not diff_states(o, e)


def test_state_digest_at_scale():
    """4096 groups x 5 peers x 1024-entry logs, lossless, to 1000 entries: digests and counters equal
    the checker's; every group agrees on every index and no safety flag is raised."""
    kw = dict(n_groups=4096, n_peers=5, log_cap=1024, seed=1, start_period=5, start_max=1000)
    o, e = _pair(**kw)
    for n in (2000, 3500):
        o.step_mt(n, 8)
        e.step(n)
        assert np.array_equal(o.state_hash(), e.state_hash())
    assert o.counters() == e.counters()
    assert e.counters()["entries_committed"] == 4096 * 1000
    assert e.safety_check() == (0, 0)
    commit = e.read("commit")
    assert (commit == 1000).all()
    assert (e.read("applyhash") == e.read("applyhash")[:, :1]).all()


def test_summary_and_one_rank_allreduce():
    """rg_safety_summary against the checker's, and rg_allreduce_summary through a one-rank RCCL communicator: the
    reduction of one rank's words is those words (counters of rg_counters, summary of rg_safety_summary)."""
    kw = dict(n_groups=700, n_peers=5, log_cap=128, seed=2, req_drop_permille=100, reply_drop_permille=100,
              start_period=5, start_max=100, part_period=100, part_jitter=20)
    o, e = _pair(**kw)
    o.step_mt(3000, 8)
    e.step(3000)
    assert o.safety_check() == e.safety_check()
    so, se = o.safety_summary(), e.safety_summary()
    assert so == se
    assert se["flagged_groups"] == e.safety_flags()[1] > 0
    bits = sum(1 << i for i, n in enumerate(_abi.FLAG_NAMES) if se["groups_" + n])
    assert bits == e.safety_flags()[0]
    assert 0 < se["rounds_flagged"] < se["rounds_all"] == int(e.read("roundseq").sum())
    assert se["recorded_all"] == int(e.read("recmax").sum())
    # no communicator: the engine's own words
    c0, s0 = e.allreduce_summary()
    assert c0 == e.counters() and s0 == se
    # a one-rank communicator made through the library (ncclCommInitRank, ncclAllReduce on the engine's stream)
    e.comm_init(0, 1)
    c1, s1 = e.allreduce_summary()
    assert c1 == e.counters() and s1 == se
    e.step(100)
    c2, s2 = e.allreduce_summary()
    assert c2 == e.counters() and c2["group_ticks"] == 700 * 3100 and s2 == e.safety_summary()
    e.comm_destroy()
    e.comm_destroy()                                  # idempotent
    assert len(type(e).comm_unique_id()) == _abi.COMM_ID_BYTES


def test_abi_from_plain_c():
    """tests/c/abi_smoke.c: the reference's first agreement test through the C ABI, no Python in between."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "abi_smoke")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
