"""Helpers shared by the parity tests: run the oracle and the HIP engine on
one config and compare every Raft-visible word."""
import numpy as np

from raft_amd import _abi

# every field of include/raftgpu.h; log arrays are compared up to each peer's length
SCALARS = [n for n, (_, k) in _abi.FIELDS.items() if k in ("p32", "p64", "m32", "g32")]


def diff_states(a, b, with_logs=True):
    """Returns a list of human-readable differences between two engines' states."""
    out = []
    for n in SCALARS:
        x, y = a.read(n), b.read(n)
        if not np.array_equal(x, y):
            idx = np.argwhere(x != y)[0]
            out.append("%s differs at %s: %s vs %s" % (n, tuple(idx), x[tuple(idx)], y[tuple(idx)]))
    if with_logs:
        for lenf, names in (("loglen", ("logterm", "logcmd")), ("snaplen", ("snaplogterm", "snaplogcmd"))):
            ln = a.read(lenf)
            L = a.L
            mask = np.arange(L)[None, None, :] < ln[:, :, None]
            for n in names:
                x, y = a.read(n), b.read(n)
                bad = (x != y) & mask
                if bad.any():
                    idx = np.argwhere(bad)[0]
                    out.append("%s differs at %s: %s vs %s" % (n, tuple(idx), x[tuple(idx)], y[tuple(idx)]))
        rm = a.read("recmax")
        mask = (np.arange(a.L)[None, :] <= rm[:, None]) & (np.arange(a.L)[None, :] >= 1)
        x, y = a.read("applyrec"), b.read("applyrec")
        bad = (x != y) & mask
        if bad.any():
            idx = np.argwhere(bad)[0]
            out.append("applyrec differs at %s" % (tuple(idx),))
    ca, cb = a.counters(), b.counters()
    for k in ca:
        if ca[k] != cb[k]:
            out.append("counter %s: %d vs %d" % (k, ca[k], cb[k]))
    return out


def run_lockstep(oracle, engine, schedule, with_logs=True, hash_too=True):
    """schedule: list of tick counts; after each chunk all state is compared."""
    t = 0
    for n in schedule:
        oracle.step(n)
        engine.step(n)
        t += n
        d = diff_states(oracle, engine, with_logs)
        assert not d, "after tick %d: %s" % (t, "; ".join(d[:6]))
        if hash_too:
            assert np.array_equal(oracle.state_hash(), engine.state_hash()), "state hash differs after tick %d" % t
    return t
