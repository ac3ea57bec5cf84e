#!/usr/bin/env python3
"""Writes tests/golden/lockstep_digests.json: counters, flags and a SHA-256 of the per-group state digests of
the CPU restatement (oracle/) on small seeded schedules.

The reference holds no golden vectors for this path and cannot run here, so these vectors are NOT outputs of
the reference: they freeze the lockstep model as the restatement computes it today, so that a change of the
model (in the restatement or in the engine) shows up as a diff of this file instead of passing silently in
both.  Regenerate deliberately:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raft_amd import _abi  # noqa: E402

CASES = {
    "boot_lossless_3": (dict(n_groups=5, n_peers=3, log_cap=64, seed=1, start_period=5, start_first_tick=20, start_max=20), [50, 150]),
    "lossless_5": (dict(n_groups=37, n_peers=5, log_cap=128, seed=3, start_period=5, start_max=100), [100, 700]),
    "lossy_partitions_5": (dict(n_groups=64, n_peers=5, log_cap=256, seed=2, req_drop_permille=100, reply_drop_permille=100,
                                start_period=5, start_max=200, part_period=100, part_jitter=20, account_bytes=1), [300, 2400]),
    "lossy_partitions_3": (dict(n_groups=64, n_peers=3, log_cap=256, seed=9, req_drop_permille=100, reply_drop_permille=100,
                                start_period=5, start_max=200, part_period=100, part_jitter=20), [300, 2400]),
    "churn_7": (dict(n_groups=48, n_peers=7, log_cap=512, seed=5, req_drop_permille=50, reply_drop_permille=50, start_period=3,
                     start_max=400, start_mode=_abi.START_ALL, churn_period=70, churn_disconnect_permille=200,
                     churn_restart_permille=500, churn_crash_permille=200), [400, 3000]),
    "reordering_heavy_loss_5": (dict(n_groups=32, n_peers=5, log_cap=256, seed=11, req_drop_permille=300, reply_drop_permille=300,
                                     reply_delay_permille=667, start_period=2, start_max=250, start_mode=_abi.START_ALL,
                                     part_period=37, part_jitter=11), [500, 1500]),
    "shard_of_many_3": (dict(n_groups=96, n_peers=3, log_cap=256, seed=5, req_drop_permille=10, reply_drop_permille=10,
                             start_period=5, start_max=200, group_base=5 * 131072 + 70000), [1000]),
}
FIELDS = ["term", "vote", "state", "commit", "applied", "loglen", "applyhash", "recmax", "gflags"]


def snapshot(engine):
    """What is frozen of one engine state: JSON-friendly, small, and sensitive to every Raft-visible word."""
    out = {"counters": engine.counters(), "safety": list(engine.safety_check()),
           "state_sha256": hashlib.sha256(engine.state_hash().tobytes()).hexdigest()}
    for f in FIELDS:
        out["sum_" + f] = int(engine.read(f).astype("uint64").sum() & 0xFFFFFFFFFFFFFFFF)
    return out


def run_case(make_engine, name):
    kw, stops = CASES[name]
    e = make_engine(_abi.make_config(**kw))
    out, t = [], 0
    for stop in stops:
        e.step(stop - t)
        t = stop
        out.append({"tick": stop, **snapshot(e)})
    return out


def main():
    from oracle.binding import OracleEngine
    golden = {name: run_case(OracleEngine, name) for name in CASES}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lockstep_digests.json")
    with open(path, "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote", path)
    # the same cases for the Go lockstep driver (bindings/go/lockstep/lockstep_test.go), which runs them on the
    # reference's own code where a Go toolchain exists
    cases = [{"name": name, "config": {k: int(v) for k, v in kw.items()}, "stops": stops} for name, (kw, stops) in sorted(CASES.items())]
    cpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lockstep_cases.json")
    with open(cpath, "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print("wrote", cpath)


if __name__ == "__main__":
    main()
