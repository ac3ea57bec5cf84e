#!/usr/bin/env python3
"""What a due launch costs before any span runs: one-tick launches over 65,536 five-peer groups whose peers are all
crashed (nothing is ever due: the kernel reads the wake words and ends), against the same with the benchmark workload live.
usage: python profiles/time_idle_due.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
G = 65536
kw = dict(n_groups=G, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100,
          start_period=5, start_max=1000, part_period=100, part_jitter=20)
for label, park in (("live", False), ("all peers crashed", True)):
    e = Engine(_abi.make_config(**kw))
    e.step(3000)
    if park:
        dead = e.read("dead")
        e.write("dead", np.ones_like(dead))
        e.step(50)                        # a full launch: every wake word now says "never"
    e.sync()
    K = 2000
    ms0 = e.step_time_ms()[0]
    t0 = time.perf_counter()
    for _ in range(K):
        e.step(1)
    e.sync()
    dt = time.perf_counter() - t0
    print("%-20s %.1f us per launch, kernel %.1f us by HIP events" % (label, dt / K * 1e6, (e.step_time_ms()[0] - ms0) / K * 1e3), flush=True)
