#!/usr/bin/env python3
"""Shader cycles per phase of a tick (diagnostic -DRG_STAMPS build).
usage: RAFTGPU_LIB=raft_amd/lib/libraftgpu_stamps.so python profiles/phase_cycles.py [idle|bench]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
G, T = int(os.environ.get("PG", 65536)), 4000
mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
kw = dict(n_groups=G, n_peers=5, log_cap=1024, seed=2)
if mode == "bench":
    kw.update(req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=1000, part_period=100, part_jitter=20)
e = Engine(_abi.make_config(**kw))
e.step(T); e.sync()
buf = (C.c_uint64 * 32)()
e._lib.rg_debug_section_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert e._lib.rg_debug_section_counts(e._h, buf) == 0
wt = ((G + 11) // 12) * T   # 12 five-peer groups per wave
tot = sum(buf[24:29])
for i, n in enumerate(["control", "classify", "actor rounds", "apply", "leader check"]):
    print("%-14s %8.1f cycles per wave-tick  (%4.1f %%)" % (n, buf[24 + i] / wt, 100.0 * buf[24 + i] / tot))
print("total          %8.1f cycles per wave-tick" % (tot / wt))
