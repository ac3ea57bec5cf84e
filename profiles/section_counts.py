#!/usr/bin/env python3
"""How often each section of k_tick runs per wave-tick on the bench workload.
Needs a -DRG_SECTION_COUNTS build: RAFTGPU_LIB=raft_amd/lib/libraftgpu_prof.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
G, T = 65536, 20000
e = Engine(_abi.make_config(n_groups=G, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100,
                            start_period=5, start_max=1000, part_period=100, part_jitter=20))
e.step(T); e.sync()
buf = (C.c_uint64 * 56)()
e._lib.rg_debug_section_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert e._lib.rg_debug_section_counts(e._h, buf) == 0
wt = G * 8 // 64 * T
names = ["pend-loop iterations", "RV sections", "monitor evals", "AE sections", "apply iterations", "start sections", "leader-check slow path"]
for i, n in enumerate(names):
    print("%-24s %14d  %.3f per wave-tick" % (n, buf[_abi.N_COUNTERS + i], buf[_abi.N_COUNTERS + i] / wt))
