#!/usr/bin/env python3
"""Shader cycles of the stretches of a span in short launches (diagnostic -DRG_DSTAMPS build, profiles/build_ab.sh dst -DRG_DSTAMPS).
usage: RAFTGPU_LIB=$PWD/raft_amd/lib/ab/libraftgpu_dst.so python profiles/due_stamps.py [ticks per launch]"""
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
spans, groups = d[49], d[50]
names = {1: "set-up: loads of the state", 2: "set-up: the rest", 3: "first look (next_event)", 4: "the passes", 5: "wind-up: client Starts",
         6: "wind-up: wake-up chains", 7: "wind-up: apply checks", 8: "wind-up: stores"}
print("%d launches of %d tick(s): %.1f spans per launch, %.2f groups per span" % (K, n, spans / K, groups / max(1, spans)))
tot = sum(d[41:49])
for k in range(1, 9):
    print("%-30s %9.0f cycles per span (%4.1f %%)" % (names[k], d[40 + k] / max(1, spans), 100.0 * d[40 + k] / max(1, tot)))
print("total %.0f cycles per span; %.2f us per span by the 100 MHz clock (shader clock %.0f MHz); spans over 60k / 120k / 200k cycles: %.4f / %.4f / %.4f" % (tot / max(1, spans), d[55] / max(1, spans) / 100.0, tot / max(1, d[55]) * 100.0, d[52] / max(1, spans), d[53] / max(1, spans), d[54] / max(1, spans)))
