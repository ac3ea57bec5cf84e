#!/usr/bin/env python3
"""Shader cycles per section of a pass in launches of one tick (diagnostic build: profiles/build_ab.sh st -DRG_STAMPS -DRG_SECTION_COUNTS).
usage: RAFTGPU_LIB=$PWD/raft_amd/lib/ab/libraftgpu_st.so python profiles/due_phases.py [ticks per launch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = int(os.environ.get("PG", 65536))
e = Engine(_abi.make_config(n_groups=G, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100,
                            start_period=5, start_max=1000, part_period=100, part_jitter=20))
e.step(3000); e.sync()
buf = (C.c_uint64 * 56)()
e._lib.rg_debug_section_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert e._lib.rg_debug_section_counts(e._h, buf) == 0
before = list(buf)
K = 400
for _ in range(K):
    e.step(n)
e.sync()
assert e._lib.rg_debug_section_counts(e._h, buf) == 0
d = [buf[i] - before[i] for i in range(56)]
NC = _abi.N_COUNTERS
passes = d[NC + 14]
names = {0: "client_starts", 1: "classify", 2: "rest of actors", 3: "apply", 4: "leader check", 5: "pend header+election", 6: "RV", 7: "monitor eval", 8: "AE",
         9: "catch_up", 10: "partition/churn (+early starts)", 11: "next_event", 12: "client starts before control"}
tot = sum(d[40 + k] for k in names)
print("%d launches of %d tick(s): %.1f passes per launch" % (K, n, passes / K))
for k, nm in names.items():
    print("%-34s %8.0f cycles per pass (%4.1f %%)" % (nm, d[40 + k] / max(1, passes), 100.0 * d[40 + k] / max(1, tot)))
print("total %.0f cycles per pass" % (tot / max(1, passes)))
extra = {0: "actor-loop iterations", 1: "RV sections", 2: "monitor evals", 3: "AE sections", 4: "apply sections", 5: "Start sections", 7: "store fences", 8: "AE: matrix cache miss",
         9: "AE: prevTerm from HBM", 10: "AE: entries from HBM", 11: "AE: backoff scan", 12: "apply: check by the wave", 13: "apply: record by the wave"}
for k, nm in extra.items():
    print("%-28s %.3f per pass" % (nm, d[NC + k] / max(1, passes)))
