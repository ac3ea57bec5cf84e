"""Shader cycles per section of one pass of the tick loop (diagnostic -DRG_STAMPS -DRG_SECTION_COUNTS build), 49,152
five-peer groups (one full batch) of the bench workload, 4,000 ticks.
usage: RAFTGPU_LIB=$PWD/raft_amd/lib/libraftgpu_stamps.so python profiles/pass_cycles.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from raft_amd import Engine, _abi
G, T = 49152, 4000
kw = dict(n_groups=G, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=1000, part_period=100, part_jitter=20)
e = Engine(_abi.make_config(**kw))
e.step(T); e.sync()
buf = (C.c_uint64 * 56)()
e._lib.rg_debug_section_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert e._lib.rg_debug_section_counts(e._h, buf) == 0
passes = buf[_abi.N_COUNTERS + 14]
names = {0: "client_starts", 1: "classify", 2: "rest of actors", 3: "apply", 4: "leader check", 5: "pend header+election", 6: "RV", 7: "monitor eval", 8: "AE", 9: "catch_up", 10: "partition/churn (+early starts)", 11: "next_event"}
tot = sum(buf[40 + k] for k in names)
for k, n in names.items():
    print("%-34s %8.0f cycles per pass (%4.1f %%)" % (n, buf[40 + k] / passes, 100.0 * buf[40 + k] / tot))
print("total %.0f cycles per pass, %d passes" % (tot / passes, passes))
