#!/usr/bin/env python3
"""bench.py -- group-ticks/s and committed entries/s of the lockstep Raft engine.

A step is one pass of the hot path over one batch of synthetic input: every
group of the shard is reset to its boot state and advanced through the whole
seeded tick schedule of the workload in one kernel launch.

Workload at every N (weak scaling, per GPU): 65,536 groups x 5 peers, 1024-entry
logs, seeded 10 % request loss + 10 % reply loss, rolling partitions every
100..119 ticks, one client Start per group every 5 ticks up to 1000 entries,
20,000 ticks.  Rank r owns global groups [r*65536, (r+1)*65536); the seeded
streams key on the global group id, so results do not depend on N.  The only
collective is one all-reduce of counters and safety summary per step, made
inside the library (rg_allreduce_summary: ncclAllReduce over xGMI).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher starts its own N rank processes (one per GPU) and prints rank 0's line.
The tick kernel is the one rg_specialize compiles for this configuration when the engine is set up (before the warm-up;
`config.kernel_build` says which kernel ran); `--generic` runs the precompiled one.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak
HBM_MEASURED_PEAK_GBS = 6290.0   # best streaming copy measured on the part


def workload_config(args, rank):
    from raft_amd import _abi
    from raft_amd.distributed import shard_base
    return _abi.make_config(
        n_groups=args.groups, n_peers=args.peers, log_cap=args.log_cap, device=int(os.environ.get("LOCAL_RANK", 0)),
        group_base=shard_base(rank, args.groups), seed=args.seed,
        req_drop_permille=args.drop, reply_drop_permille=args.drop,
        start_period=5, start_first_tick=0, start_max=args.entries,
        part_period=args.part_period, part_jitter=20 if args.part_period else 0)


def algorithmic_bytes(c, P):
    """Bytes the lockstep model has to move, from the engine's event counters, as (floor, events): the floor is
    the 40 B scalar block of every peer once per tick; per event: a vote request 64 B, an AppendEntries 124 B +
    24 B per entry, one commit scan (4P+4 B) per replication round, 24 B per applied entry, 12 B per persist
    header and per newly persisted entry, 12 B per appended entry."""
    rv_delivered = c["rpc_delivered"] - c["ae_delivered"]
    floor = c["group_ticks"] * 40 * P
    events = (rv_delivered * 64 + c["ae_delivered"] * 124 + c["entries_replicated"] * 24 +
              c["ae_rounds"] * (4 * P + 4) + c["entries_applied"] * 24 + c["persists"] * 12 + c["persist_entries"] * 12 +
              c["entries_started"] * 12)
    return floor, events


def launch_ranks(n, script=None, argv=None, check_devices=True):
    """`--gpus N` without a launcher: N fresh rank processes of this script, one per GPU, started before anything
    here has touched a GPU; rank 0's output is passed on.  Returns the exit status for this process."""
    import socket
    import subprocess
    if check_devices:
        import torch
        have = torch.cuda.device_count()          # counts devices without initialising one
        if have < n:
            raise SystemExit("--gpus %d: this host shows %d GPU(s)" % (n, have))
    script = script or os.path.abspath(__file__)
    argv = sys.argv[1:] if argv is None else argv
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's pipe is read on a thread of its own, so that this loop can watch every child: a rank that dies early
    # would otherwise leave the others waiting in their rendezvous or a collective until its timeout
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if bad:
            failed = abs(bad[0])
            for p in procs:                       # fresh children of this process: no re-exec, plain termination
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.05)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return failed or max(abs(rc) for rc in rcs)


def cpu_baseline(args, eng):
    """The CPU checker (a port of the reference's logic to the same lockstep
    schedule) timed on the host cores, on a bounded sample of the workload; its
    result for those groups is then held against the engine's last step."""
    from oracle.binding import OracleEngine
    from raft_amd import _abi
    threads = max(1, min(16, os.cpu_count() or 1))
    groups = args.cpu_groups
    cfg = workload_config(args, 0)
    cfg.n_groups = groups
    o = OracleEngine(cfg)
    t0 = time.time()
    o.step_mt(args.ticks, threads)
    dt = time.time() - t0
    c = o.counters()
    o1 = OracleEngine(cfg)
    one_groups = max(1, groups // threads)
    cfg1 = workload_config(args, 0)
    cfg1.n_groups = one_groups
    o1 = OracleEngine(cfg1)
    t0 = time.time()
    o1.step(args.ticks)
    dt1 = time.time() - t0
    import numpy as np
    n = min(groups, args.groups)
    same_state = bool(np.array_equal(o.state_hash()[:n], eng.state_hash()[:n]))
    same_flags = bool(np.array_equal(o.read("gflags")[:n], eng.read("gflags")[:n]))
    return {
        "check": {"groups": n, "state_digests_equal": same_state, "safety_flags_equal": same_flags,
                  "note": "per-group digest of every Raft-visible word after the step, engine vs the port; the flags are raised "
                          "by the reference's own logic under partitions (the port raises the same ones)"},
        "value": c["group_ticks"] / dt, "unit": "group-ticks/s", "cores": threads, "kind": "port",
        "sample": "first %d groups of the workload x %d ticks, groups sharded over %d threads" % (groups, args.ticks, threads),
        "committed_entries_per_s": c["entries_committed"] / dt,
        "single_thread_value": one_groups * args.ticks / dt1,
        "go_reference": "not run: no Go toolchain on this host",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--groups", type=int, default=65536, help="groups per GPU")
    ap.add_argument("--peers", type=int, default=5)
    ap.add_argument("--log-cap", type=int, default=1024)
    ap.add_argument("--entries", type=int, default=1000)
    ap.add_argument("--ticks", type=int, default=20000)
    ap.add_argument("--drop", type=int, default=100, help="request and reply loss, permille")
    ap.add_argument("--part-period", type=int, default=100)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--launch-ticks", type=int, default=0, help="ticks per kernel launch (0 = the whole schedule)")
    ap.add_argument("--cpu-groups", type=int, default=16384)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--generic", action="store_true",
                    help="run the precompiled generic kernel instead of the one rg_specialize compiles for this configuration")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="run the N ranks on GPU 0 over gloo: checks the multi-rank path on a 1-GPU box; its line is marked and is no measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, check_devices=not args.rehearse_on_one_gpu))

    import torch
    import torch.distributed as dist
    from raft_amd import Engine, _abi

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    red_dev = torch.device("cpu") if args.rehearse_on_one_gpu else torch.device("cuda", local)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    cfg = workload_config(args, rank)
    cfg.device = local
    eng = Engine(cfg)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    specialized = False
    if not args.generic:
        try:
            eng.specialize()            # the tick kernel compiled for this configuration (about a second, before the warm-up)
            specialized = True
        except Exception as err:      # hiprtc missing: the generic kernel runs, and the line says so
            sys.stderr.write("bench: rg_specialize failed (%s); running the generic kernel\n" % err)
    launch = args.launch_ticks or args.ticks
    from raft_amd.distributed import all_reduce_words, flags_of, join_comm
    # The job's one collective: rg_allreduce_summary (ncclAllReduce inside the library, on the engine's stream) over a
    # communicator of all ranks; torch.distributed only carries the 128-byte id to the ranks.  Ranks that cannot form an
    # RCCL communicator (the one-GPU rehearsal) sum the same packed words over torch.distributed instead.
    in_library = world > 1 and not args.rehearse_on_one_gpu and join_comm(eng, red_dev)
    collective = ("none (one rank)" if world == 1 else
                  "rg_allreduce_summary: one ncclAllReduce (RCCL) of %d uint64 per step, in the library" % (_abi.N_COUNTERS + _abi.N_SUMMARY) if in_library else
                  "torch.distributed all_reduce (%s) of the library's packed words" % dist.get_backend())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        eng.reset()
        done = 0
        while done < args.ticks:
            n = min(launch, args.ticks - done)
            eng.step(n)
            done += n
        # counters and safety summary: the only cross-GPU exchange of the path
        words = eng.allreduce_summary()              # reduced over the ranks when the engine has a communicator
        return words if in_library else all_reduce_words(*words, red_dev)

    for _ in range(args.warmup):
        one_step()
    ms0, n0 = eng.step_time_ms()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tot = one_step()
    barrier()
    dt = time.perf_counter() - t0
    ms1, n1 = eng.step_time_ms()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    counters, summary = tot
    flags, flagged_groups = flags_of(summary)
    group_ticks_per_step = world * args.groups * args.ticks

    # roofline of the tick kernel on this rank: algorithmic bytes of one launch
    # over its mean duration (HIP events on the launch stream, inside the engine)
    local_c = eng.counters()
    launches = max(1, n1 - n0)
    kernel_ms = (ms1 - ms0) / launches
    launches_per_step = (args.ticks + launch - 1) // launch
    floor_bytes, event_bytes = (b / launches_per_step for b in algorithmic_bytes(local_c, args.peers))
    bytes_per_launch = floor_bytes + event_bytes
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch from hardware counters cannot be read from inside this process: they come from separate
    # rocprofv3 --pmc passes over this same command (profiles/run_pmc.sh), stored next to the profiles.
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and launch == 20000 and args.groups == 65536 and args.peers == 5:   # collected on the default launch only
        try:
            rec = json.load(open(pmc))
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_source = "profiles/pmc_traffic.json (%s): not measured in this run" % rec.get("collected", "rocprofv3 --pmc passes of this command")
        except Exception:
            traffic = None

    # What actually bounds the fused kernel is instruction issue, not memory: VALU wave-instructions per launch (from a
    # rocprofv3 --pmc pass of this same command, like the traffic) x 4 cycles each on one of 1,024 SIMDs, at the shader
    # clock measured in that pass, over this run's kernel time.
    issue = None
    pmi = os.path.join(ROOT, "profiles", "pmc_issue.json")
    if os.path.exists(pmi) and launch == 20000 and args.groups == 65536 and args.peers == 5 and specialized:
        try:
            rec = json.load(open(pmi))
            valu, mhz = float(rec["SQ_INSTS_VALU_per_launch"]), float(rec["shader_clock_mhz"])
            floor_ms = valu * 4.0 / (1024 * mhz * 1e6) * 1e3
            issue = {"valu_wave_insts_per_launch": valu, "cycles_per_inst": 4, "simds": 1024, "shader_clock_mhz": mhz,
                     "issue_floor_ms": floor_ms, "frac": floor_ms / kernel_ms,
                     "valu_busy_frac_of_simd_cycles": rec.get("valu_busy_frac"), "exec_mask_occupancy": rec.get("exec_mask_occupancy"),
                     "source": "profiles/pmc_issue.json (%s): counters not measured in this run, kernel_ms is" % rec.get("collected", "rocprofv3 --pmc passes of this command")}
        except Exception:
            issue = None
    fused = launch > 1

    out = {
        "metric": "group_ticks_per_sec", "value": group_ticks_per_step * args.steps / dt, "unit": "group-ticks/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "committed_entries_per_sec": counters["entries_committed"] * args.steps / dt,
        "config": {
            "workload": "%d groups/GPU x %d peers, %d-entry logs, %d/1000 request + %d/1000 reply loss, rolling partitions "
                        "every %d..%d ticks, Start every 5 ticks to %d entries, %d ticks/step" % (
                            args.groups, args.peers, args.log_cap, args.drop, args.drop, args.part_period,
                            args.part_period + 19, args.entries, args.ticks),
            "groups_total": world * args.groups, "peers": args.peers, "ticks_per_step": args.ticks,
            "ticks_per_launch": launch, "parallelism": "groups sharded x%d, no data-path collective" % world, "seed": args.seed,
            "collective": collective,
            "kernel_build": "compiled at engine set-up for this configuration (rg_specialize)" if specialized else "generic, precompiled",
        },
        "roofline": {
            # achieved / peak / frac: the byte model (algorithmic bytes per launch over the kernel's time against HBM
            # peak).  In fused mode 85 % of those bytes -- the per-tick floor -- never move, so the model can exceed
            # what a streaming kernel reaches and does not name the limiter: `bound` does (VALU issue, see `issue`).
            "bound": "valu_issue" if fused else "launch_latency", "model_bound": "hbm", "issue": issue,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "k_tick_spec" if specialized else "k_tick", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
            # the two parts of the algorithmic figure: the per-tick floor (40 B x peers x group-ticks), which a fused
            # launch never moves, and the bytes of the counted events; and the same rate against the measured peak
            "floor_bytes": floor_bytes, "event_bytes": event_bytes,
            "frac_event_bytes": event_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_traffic": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "measured_peak": HBM_MEASURED_PEAK_GBS, "frac_of_measured_peak": achieved / HBM_MEASURED_PEAK_GBS,
            "mode": ("fused: peer scalars stay in registers and LDS within a launch of %d ticks (cut into relay chunks of "
                     "at least 1,000 ticks, at most 12, when the groups exceed what the chip holds at once)" % launch) if launch > 1
                    else "streamed: one launch per tick; a launch runs only the groups with an event in that tick (due launch), the others' state stays in HBM untouched; kernel_ms: one launch in 16 is timed (its two event packets included, so it lies above the wall time per launch)",
        },
        "safety_flags": flags, "flagged_groups": flagged_groups,
        # the flags are raised by the reference's own rules under partitions (tests/test_repaired_rules.py: with its vote
        # rule, persisted prefix and `restart` repaired the same schedule ends with none); per bit, and how much of the
        # run's work happened in flagged groups
        "safety": {
            "flagged_groups_by_bit": {n: summary["groups_" + n] for n in _abi.FLAG_NAMES},
            "flagged_groups": flagged_groups, "groups_total": world * args.groups,
            "share_of_groups": flagged_groups / float(world * args.groups),
            "share_of_rounds_sent": summary["rounds_flagged"] / float(max(1, summary["rounds_all"])),
            "share_of_entries_recorded": summary["recorded_flagged"] / float(max(1, summary["recorded_all"])),
            "note": "rounds = RequestVote + AppendEntries rounds (each up to peers-1 Calls); entries recorded = highest index a "
                    "group's tester saw applied; a group counts as flagged from the tick its first flag rises to the end of the run",
        },
        "counters_per_step": counters,
    }
    if args.rehearse_on_one_gpu:
        out["rehearsal"] = "all %d ranks shared GPU 0 over gloo: not a measurement" % world
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args, eng)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
