#!/usr/bin/env python3
"""Time of short launches on the benchmark workload: due launches (only the groups with an event inside the launch run)
against full launches of the same length.
usage: python profiles/time_short_launches.py [groups]      (WINS="8,16,32": windows to try; NS="1,5": launch lengths)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
G = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kw = dict(n_groups=G, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100,
          start_period=5, start_max=1000, part_period=100, part_jitter=20)
ns = [int(x) for x in os.environ.get("NS", "1,2,5,8,16,32").split(",")]
wins = os.environ.get("WINS", "").split(",")
span = int(os.environ.get("SPAN", "2000"))       # ticks timed per setting


def run(label, env):
    for k in ("RAFTGPU_DUE_MAX_TICKS", "RAFTGPU_DUE_WIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(_abi.make_config(**kw))
    if os.environ.get("SPEC"):
        e.specialize()                   # the kernels compiled for this configuration
    out = []
    for n in ns:
        e.reset(); e.step(3000); e.sync()
        k = max(1, span // n)
        ms0 = e.step_time_ms()[0]
        t0 = time.perf_counter()
        for _ in range(k):
            e.step(n)
        e.sync()
        dt = time.perf_counter() - t0
        ev = (e.step_time_ms()[0] - ms0) / k * 1e3       # HIP events around each launch, on its stream
        out.append("n=%d: %.1f us/launch (kernel %.1f) %.2e gt/s" % (n, dt / k * 1e6, ev, G * n * k / dt))
    h = int(e.state_hash().sum() & 0xffffffff)
    print("%-22s %s  digest %08x" % (label, " | ".join(out), h), flush=True)
    del e


run("full launches", {"RAFTGPU_DUE_MAX_TICKS": "0"})
run("due launches", {"RAFTGPU_DUE_MAX_TICKS": "1000"})
for w in wins:
    if w:
        run("due, window %s" % w, {"RAFTGPU_DUE_MAX_TICKS": "1000", "RAFTGPU_DUE_WIN": w})
