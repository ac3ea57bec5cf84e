#!/usr/bin/env python3
"""First tick at which one group differs between the oracle and the engine, with the words that differ and the
state one tick before.  usage: python profiles/first_diff.py <global group id> [ticks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raft_amd import Engine, _abi
from oracle.binding import OracleEngine
from tests.parity import diff_states
CFG3 = dict(n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5,
            start_max=1000, part_period=100, part_jitter=20)
gid, T = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 20000
step = int(os.environ.get("STEP", 1))
o = OracleEngine(_abi.make_config(n_groups=1, group_base=gid, **CFG3))
e = Engine(_abi.make_config(n_groups=1, group_base=gid, **CFG3))
F = ("term", "vote", "state", "commit", "applied", "loglen", "timer", "timeout", "hbtimer", "elnext", "monnext", "rngctr", "roundseq",
     "stop", "cnt", "restart", "first", "monalive", "connected", "side", "snaplen")
import ctypes as C
def caches(eng):
    out = []
    for fid in (80, 81, 82):
        a = np.empty((1, 5), dtype=np.int32)
        assert eng._lib.rg_read_state(eng._h, fid, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
        out.append(a[0].tolist())
    return out
def cache_check(eng, t):
    last, first, cterm = caches(eng)
    lt, ln, cm, tm = eng.read("logterm")[0], eng.read("loglen")[0], eng.read("commit")[0], eng.read("term")[0]
    for p in range(5):
        if last[p] != lt[p, ln[p] - 1] or cterm[p] != lt[p, min(cm[p], 1023)]:
            print("stale cache after tick", t, "peer", p, "lastTerm", last[p], lt[p, ln[p] - 1], "commitTerm", cterm[p], lt[p, cm[p]], "commit", cm[p], "loglen", ln[p])
            return True
    return False
prev = None
stale = False
for t in range(0, T, step):
    snap = {f: o.read(f)[0].tolist() for f in F}
    snap["next"], snap["match"] = o.read("nextindex")[0].tolist(), o.read("matchindex")[0].tolist()
    o.step(step); e.step(step)
    if not stale and os.environ.get("CACHES", "1") == "1" and cache_check(e, t):
        stale = True
        print("state before that tick:")
        for k, v in snap.items():
            print("  ", k, v)
        print("engine after:", {f: e.read(f)[0].tolist() for f in F})
    if not np.array_equal(o.state_hash(), e.state_hash()):
        print("first difference after tick", t, "(launch of %d)" % step)
        for d in diff_states(o, e)[:20]:
            print("  ", d)
        print("state before that tick (both):")
        for k, v in snap.items():
            print("  ", k, v)
        print("oracle after:", {f: o.read(f)[0].tolist() for f in F})
        print("engine after:", {f: e.read(f)[0].tolist() for f in F})
        print("counters oracle - engine:", {k: v - e.counters()[k] for k, v in o.counters().items() if v != e.counters()[k]})
        break
else:
    print("no difference in", T, "ticks")
