"""ctypes handle on the CPU checker (oracle/libraft_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and
by bench.py's cpu_baseline leg.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

from raft_amd import _abi
from raft_amd._abi import AppendArgs, AppendReply, ApplyMsgC, VoteArgs, VoteReply, Config
from raft_amd.engine import EngineBase, RaftGpuError

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libraft_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "raft_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "libraft_oracle.so"], stdout=subprocess.DEVNULL)


def load(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None and os.path.isdir("/root/reference") is False and not os.path.exists(LIB):
        raise RuntimeError("oracle library missing: " + LIB)
    if path is None and not os.path.exists(LIB):
        build()
    lib = C.CDLL(path or LIB)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    P = C.POINTER
    sig = {
        "ro_create": ([P(Config)], vp), "ro_destroy": ([vp], None), "ro_reset": ([vp], None),
        "ro_set_full_persist": ([vp, C.c_int], None),
        "ro_step": ([vp, i32], C.c_int), "ro_step_mt": ([vp, i32, i32], C.c_int), "ro_now": ([vp], i64),
        "ro_start": ([vp, i32, i32, C.c_uint64, P(i32), P(i32), P(i32)], C.c_int),
        "ro_get_state": ([vp, i32, i32, P(i32), P(i32)], C.c_int),
        "ro_set_connected": ([vp, i32, i32, i32], C.c_int), "ro_set_side": ([vp, i32, i32, i32], C.c_int),
        "ro_crash": ([vp, i32, i32], C.c_int), "ro_restart": ([vp, i32, i32], C.c_int), "ro_kill": ([vp, i32, i32], C.c_int),
        "ro_set_repaired_rules": ([vp, C.c_int], None), "ro_safety_summary": ([vp, P(C.c_uint64)], C.c_int),
        "ro_set_unreliable": ([vp, i32, i32, i32], C.c_int),
        "ro_set_unreliable_group": ([vp, i32, i32, i32, i32], C.c_int),
        "ro_reset_groups": ([vp, P(i32), i64], C.c_int),
        "ro_set_schedule": ([vp, P(_abi.SchedOp), i64], C.c_int),
        "ro_rpc_request_vote": ([vp, i32, i32, P(VoteArgs), P(VoteReply)], C.c_int),
        "ro_rpc_append_entries": ([vp, i32, i32, P(AppendArgs), P(i32), P(C.c_uint64), P(AppendReply)], C.c_int),
        "ro_read_state": ([vp, i32, vp, i64], C.c_int), "ro_write_state": ([vp, i32, vp, i64], C.c_int),
        "ro_state_hash": ([vp, P(C.c_uint64), i64], C.c_int),
        "ro_counters": ([vp, P(C.c_uint64)], C.c_int),
        "ro_rpc_bytes": ([vp, C.c_int64, P(C.c_uint64)], C.c_int),
        "ro_safety_check": ([vp], C.c_int), "ro_safety_flags": ([vp, P(C.c_uint32)], C.c_int),
        "ro_snapshot_export": ([vp, i32, i32, vp, i64, P(i64)], C.c_int),
        "ro_snapshot_import": ([vp, i32, i32, vp, i64], C.c_int),
        "ro_drain_applied": ([vp, i32, i32, P(ApplyMsgC), i64, P(i64)], C.c_int),
        "ro_leader_backoff": ([vp, i32, i32, i32, i32, i32], i32),
        "ro_commit_rule": ([vp, i32, i32], i32),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    if path is None:
        _lib = lib
    return lib


class OracleEngine(EngineBase):
    """Same Python surface as raft_amd.Engine, served by the CPU restatement."""

    def __init__(self, cfg=None, lib_path=None, **kw):
        self._lib = load(lib_path)
        self.cfg = cfg if cfg is not None else _abi.make_config(**kw)
        self.G, self.P, self.L = self.cfg.n_groups, self.cfg.n_peers, self.cfg.log_cap
        self._h = self._lib.ro_create(C.byref(self.cfg))
        if not self._h:
            raise ValueError("ro_create rejected the config")

    def _fn(self, name):
        return getattr(self._lib, "ro_" + name)

    def _rw(self, name, field, ptr, nbytes):
        return self._fn(name)(self._h, field, ptr, nbytes)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ro_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        self._lib.ro_reset(self._h)

    def sync(self):
        pass

    def now(self):
        return self._lib.ro_now(self._h)

    def set_full_persist(self, on):
        self._lib.ro_set_full_persist(self._h, 1 if on else 0)

    def set_repaired_rules(self, mask=7):
        """TEST-ONLY: repair the reference's rules (bit 1: vote on the last log entry, 2: persist the whole log,
        4: a demoting handler clears `restart`), to tell its own violations from a model defect."""
        self._lib.ro_set_repaired_rules(self._h, int(mask))

    def step_mt(self, n_ticks, n_threads):
        self._check(self._lib.ro_step_mt(self._h, n_ticks, n_threads), "step_mt")

    def request_vote(self, g, p, term, candidate, last_log_index, last_log_term):
        a, r = VoteArgs(term, candidate, last_log_index, last_log_term), VoteReply(0, 0)
        self._check(self._lib.ro_rpc_request_vote(self._h, g, p, C.byref(a), C.byref(r)), "rpc_request_vote")
        return r.term, bool(r.granted)

    def request_vote_batch(self, groups, peers, args):
        return [self.request_vote(g, p, *a) for g, p, a in zip(groups, peers, args)]

    def append_entries(self, g, p, term, leader, prev_log_index, prev_log_term, entries, leader_commit):
        n = len(entries)
        a = AppendArgs(term, leader, prev_log_index, prev_log_term, leader_commit, n, 0)
        r = AppendReply(0, 0, 0, 0)
        ta = (C.c_int32 * max(1, n))(*[t for t, _ in entries])
        ca = (C.c_uint64 * max(1, n))(*[c for _, c in entries])
        self._check(self._lib.ro_rpc_append_entries(self._h, g, p, C.byref(a), ta, ca, C.byref(r)), "rpc_append_entries")
        return r.term, bool(r.success), r.conflict_term, r.conflict_first_index

    def append_entries_batch(self, groups, peers, reqs):
        return [self.append_entries(g, p, *q) for g, p, q in zip(groups, peers, reqs)]

    def set_unreliable_group(self, g, req_drop=100, reply_drop=100, reply_delay=0):
        self._check(self._lib.ro_set_unreliable_group(self._h, g, req_drop, reply_drop, reply_delay), "set_unreliable_group")

    def reset_groups(self, groups):
        a = (C.c_int32 * len(groups))(*[int(g) for g in groups])
        self._check(self._lib.ro_reset_groups(self._h, a, len(groups)), "reset_groups")

    def start_batch(self, groups, peers, cmds):
        return [self.start(int(g), int(p), int(c)) for g, p, c in zip(groups, peers, cmds)]

    def leader_backoff(self, g, leader, follower, conflict_term, conflict_first_index):
        return self._lib.ro_leader_backoff(self._h, g, leader, follower, conflict_term, conflict_first_index)

    def commit_rule(self, g, leader):
        return self._lib.ro_commit_rule(self._h, g, leader)

    def drain_applied(self, g, p, cap=None):
        """applyCh: the ApplyMsgs the slot emitted since the last drain, as emitted (needs apply_ring_cap > 0)."""
        cap = cap or max(self.L, self.cfg.apply_ring_cap)
        buf = (ApplyMsgC * cap)()
        n = C.c_int64()
        st = self._lib.ro_drain_applied(self._h, g, p, buf, cap, C.byref(n))
        out = [(int(buf[i].command_index), int(buf[i].command)) for i in range(n.value)]
        if st != 0:
            err = RaftGpuError(st, "drain_applied")
            err.partial = out
            raise err
        return out
