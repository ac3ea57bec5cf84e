#!/usr/bin/env python3
"""Kernel time of the benchmark schedule at a chosen group count, for A/B builds.
usage: RAFTGPU_LIB=<lib> PG=<groups> python profiles/time_ticks.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
T = 20000
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("RAFTGPU_LIB", "default"))
for G in [int(x) for x in os.environ.get("PG", "65536").split(",")]:
    e = Engine(_abi.make_config(n_groups=G, n_peers=5, log_cap=1024, seed=1, req_drop_permille=100, reply_drop_permille=100,
                                start_period=5, start_max=1000, part_period=100, part_jitter=20))
    best, last = 1e30, 0.0
    for rep in range(2):
        e.reset(); e.step(T); e.sync()
        tot = e.step_time_ms()[0]
        best, last = min(best, tot - last), tot
    c = e.counters()
    print("%s groups %d  %.2f ms  %.3e group-ticks/s  committed %d" % (tag, G, best, G * T / best * 1e3, c["entries_committed"]), flush=True)
    del e
