#!/usr/bin/env python3
"""Kernel time of the other named configurations (one rank's share), for the numbers table.
usage: python profiles/time_configs.py       (GENERIC=1: the precompiled kernel instead of the one rg_specialize builds;
ONLY=cfg4: one configuration; LAYOUT=1 rows / 2 columns: the log layout instead of the one the engine picks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
CASES = {
    "cfg2 4096x5 lossless, L=1024, 10000 ticks": (dict(n_groups=4096, n_peers=5, log_cap=1024, seed=1, start_period=5, start_max=1000), 10000),
    "cfg3 65536x5 lossy+partitions, L=1024, 20000 ticks": (dict(n_groups=65536, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=1000, part_period=100, part_jitter=20), 20000),
    "target 131072x5 (one of 8 shards of 1M five-peer groups) lossy+partitions, L=1024, 20000 ticks": (dict(n_groups=131072, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=1000, part_period=100, part_jitter=20, group_base=5 * 131072), 20000),
    "cfg4 16384x7 churn, L=4096, 40000 ticks": (dict(n_groups=16384, n_peers=7, log_cap=4096, seed=4, req_drop_permille=50, reply_drop_permille=50, start_period=10, start_max=3500, start_mode=_abi.START_ALL, churn_period=70, churn_disconnect_permille=200, churn_restart_permille=500, churn_crash_permille=200), 40000),
    "cfg5 131072x3 (one of 8 shards) 1% loss, L=256, 4000 ticks": (dict(n_groups=131072, n_peers=3, log_cap=256, seed=5, req_drop_permille=10, reply_drop_permille=10, start_period=5, start_max=200, group_base=5 * 131072), 4000),
}
only = os.environ.get("ONLY")          # e.g. ONLY=cfg4
for name, (kw, T) in CASES.items():
    if only and not name.startswith(only):
        continue
    if os.environ.get("NGROUPS"):           # tuning aid: the same configuration with another number of groups
        kw = dict(kw, n_groups=int(os.environ["NGROUPS"]))
    if os.environ.get("LAYOUT"):
        kw = dict(kw, log_layout=int(os.environ["LAYOUT"]))
        name += " [layout %s]" % {"1": "rows", "2": "columns"}.get(os.environ["LAYOUT"], "auto")
    e = Engine(_abi.make_config(**kw))
    if not os.environ.get("GENERIC"):
        e.specialize()
    best, last = 1e30, 0.0
    for rep in range(2):
        e.reset(); e.step(T); e.sync()
        tot = e.step_time_ms()[0]
        best, last = min(best, tot - last), tot
    c = e.counters()
    G = kw["n_groups"]
    print("%-98s %8.2f ms  %.3e group-ticks/s  %.3e committed entries/s  flags %s" % (name, best, G * T / best * 1e3, c["entries_committed"] / best * 1e3, e.safety_flags()), flush=True)
    del e
