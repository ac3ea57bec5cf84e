#!/usr/bin/env python3
"""Parity triage on the benchmark workload: which groups of a slice differ from the oracle, in a plain launch of the
slice alone and inside the full-size relay launch.  usage: python profiles/bisect_cfg3.py [first] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raft_amd import Engine, _abi
from oracle.binding import OracleEngine
CFG3 = dict(n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5,
            start_max=1000, part_period=100, part_jitter=20)
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 24576, int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = int(os.environ.get("PT", 20000))
o = OracleEngine(_abi.make_config(n_groups=count, group_base=first, **CFG3)); o.step_mt(T, 16); ho = o.state_hash()
e = Engine(_abi.make_config(n_groups=count, group_base=first, **CFG3)); e.step(T); he = e.state_hash()
bad = np.nonzero(ho != he)[0]
print("plain launch of the slice: %d groups differ %s" % (len(bad), (bad[:8] + first).tolist()), flush=True)
del e
if os.environ.get("FULL", "1") == "1":
    e = Engine(_abi.make_config(n_groups=65536, **CFG3)); e.step(T); he = e.state_hash()[first:first + count]
    bad = np.nonzero(ho != he)[0]
    print("inside the full relay launch: %d groups differ %s" % (len(bad), (bad[:8] + first).tolist()), flush=True)
    if len(bad):
        g = int(bad[0])
        for f in ("term", "vote", "state", "commit", "applied", "loglen", "timer", "timeout", "hbtimer", "elnext", "monnext", "rngctr", "roundseq", "snaplen", "side", "nextindex", "matchindex"):
            a, b = o.read(f)[g], e.read(f)[first + g]
            if not np.array_equal(a, b):
                print(f, a.tolist(), b.tolist())
        print("group words", [(f, int(o.read(f)[g]), int(e.read(f)[first + g])) for f in ("startseq", "partnext", "partepoch", "recmax", "gflags")])
