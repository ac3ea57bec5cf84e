#!/usr/bin/env python3
"""Host-driven use of the engine, the way the reference's tester drives a Raft instance: the client Starts come from the
host (rg_start_batch: host buffers in, results out over PCIe), five ticks per launch in between.
usage: python profiles/time_host_driven.py [groups]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
G = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
e = Engine(_abi.make_config(n_groups=G, n_peers=5, log_cap=1024, seed=1))
e.step(40)                                   # peer 0 of every group wins the first election on a lossless net (raft.go:545)
dt_req = np.dtype([("group", "<i4"), ("peer", "<i4"), ("command", "<u8"), ("index", "<i4"), ("term", "<i4"), ("is_leader", "<i4"), ("pad", "<i4")])
reqs = np.zeros(G, dtype=dt_req)             # the C ABI's rg_start_req array, filled without a Python loop
reqs["group"] = np.arange(G)
import ctypes as C
ptr = reqs.ctypes.data_as(C.POINTER(_abi.StartReq))
rounds = 100
e.sync()
t0 = time.perf_counter()
accepted = 0
parts = [0.0, 0.0, 0.0, 0.0]              # filling the requests, rg_start_batch, counting the answers, rg_step
for i in range(rounds):
    ta = time.perf_counter()
    reqs["command"] = (np.arange(G, dtype=np.uint64) << np.uint64(20)) + np.uint64(i + 1)
    tb = time.perf_counter()
    e._check(e._lib.rg_start_batch(e._h, ptr, G), "start_batch")   # what Engine.start_batch_arrays does, without its copies
    tc = time.perf_counter()
    accepted += int(np.count_nonzero(reqs["is_leader"]))
    td = time.perf_counter()
    e.step(5)
    te = time.perf_counter()
    for k, v in enumerate((tb - ta, tc - tb, td - tc, te - td)):
        parts[k] += v
e.sync()
dt = time.perf_counter() - t0
print("per round: fill %.0f us, rg_start_batch %.0f us, count %.0f us, rg_step (asynchronous) %.0f us" % tuple(x / rounds * 1e6 for x in parts))
c = e.counters()
print("%d groups: %d rounds of (Start on every group + 5 ticks) in %.3f s: %.3e Starts/s (%d accepted), %.3e group-ticks/s, %d entries committed, flags %s" % (
    G, rounds, dt, G * rounds / dt, accepted, G * 5 * rounds / dt, c["entries_committed"], e.safety_flags()))
