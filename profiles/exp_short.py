#!/usr/bin/env python3
"""One-tick launches under different loads: every group crashed (the launch overhead), an idle lossless net, the benchmark
workload at several window sizes and group counts (wall time per launch, and HIP events around each launch).
usage: python profiles/exp_short.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAFTGPU_DUE_MAX_TICKS"] = "1000"
from raft_amd import Engine, _abi
import numpy as np
def timeit(e, n, k):
    e.sync(); ms0 = e.step_time_ms()[0]
    t0 = time.perf_counter()
    for _ in range(k): e.step(n)
    e.sync(); dt = time.perf_counter() - t0
    return dt / k * 1e6, (e.step_time_ms()[0] - ms0) / k * 1e3
bench = dict(n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=1000, part_period=100, part_jitter=20)
for label, G, kw, win, prep in [
    ("A all crashed 65536", 65536, dict(n_peers=5, log_cap=1024, seed=2), None, "crash"),
    ("B idle lossless 65536", 65536, dict(n_peers=5, log_cap=1024, seed=2), None, None),
    ("C bench 65536 win16", 65536, bench, None, None),
    ("C bench 65536 win64", 65536, bench, 64, None),
    ("D bench 4096 win64", 4096, bench, 64, None),
    ("D bench 4096 win1", 4096, bench, 1, None),
    ("D bench 256 win1", 256, bench, 1, None),
]:
    if win: os.environ["RAFTGPU_DUE_WIN"] = str(win)
    else: os.environ.pop("RAFTGPU_DUE_WIN", None)
    e = Engine(_abi.make_config(n_groups=G, **kw))
    if prep == "crash":
        z = np.ones((G, kw["n_peers"]), dtype=np.int32)
        e.write("dead", z)
        e.step(100)
    else:
        e.step(3000)
    w, k = timeit(e, 1, 500)
    print("%-28s 1-tick launch: wall %.1f us, events %.1f us" % (label, w, k), flush=True)
    del e
