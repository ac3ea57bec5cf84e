"""Engine: Python handle on the HIP engine through the C ABI (include/raftgpu.h).

One Engine owns n_groups x n_peers Raft peers resident in HBM.  The methods
are the reference's verbs in batched form; `raft_amd.raft` wraps one
(group, peer) pair in the reference's own `Make / Start / GetState / Kill`
surface (reference: src/raft/raft.go:700,344,90,377).
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (AppendArgs, AppendReply, ApplyMsgC, Config, StartReq, VoteArgs, VoteReply, COUNTERS, FIELDS,
                   N_COUNTERS, N_SUMMARY, SUMMARY)


class RaftGpuError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        RuntimeError.__init__(self, "%s failed: status %d" % (what, status))


def field_shape(name, G, P, L):
    kind = FIELDS[name][1]
    return {"p32": ((G, P), np.int32), "p64": ((G, P), np.uint64), "m32": ((G, P, P), np.int32),
            "l32": ((G, P, L), np.int32), "l64": ((G, P, L), np.uint64), "g32": ((G,), np.int32),
            "r64": ((G, L), np.uint64)}[kind]


class EngineBase(object):
    """Shared numpy plumbing over a C library exporting the rg_-shaped calls."""

    G = P = L = 0

    # subclasses provide _call(name, *args) -> status and self._h
    def _check(self, st, what):
        if st != 0:
            raise RaftGpuError(st, what)

    def read(self, name):
        shape, dt = field_shape(name, self.G, self.P, self.L)
        out = np.empty(shape, dtype=dt)
        self._check(self._rw("read_state", FIELDS[name][0], out.ctypes.data_as(C.c_void_p), out.nbytes), "read_state(%s)" % name)
        return out

    def write(self, name, arr):
        shape, dt = field_shape(name, self.G, self.P, self.L)
        a = np.ascontiguousarray(arr, dtype=dt).reshape(shape)
        self._check(self._rw("write_state", FIELDS[name][0], a.ctypes.data_as(C.c_void_p), a.nbytes), "write_state(%s)" % name)

    def dump(self, names=None):
        return {n: self.read(n) for n in (names or FIELDS.keys())}

    def counters(self):
        buf = (C.c_uint64 * N_COUNTERS)()
        self._check(self._fn("counters")(self._h, buf), "counters")
        return {n: int(buf[i]) for i, n in enumerate(COUNTERS)}

    def rpc_bytes(self, command_bytes=8):
        """cfg.bytesTotal(), src/raft/config.go:292-294, under the fixed-width size model of include/raftgpu.h."""
        out = C.c_uint64()
        self._check(self._fn("rpc_bytes")(self._h, int(command_bytes), C.byref(out)), "rpc_bytes")
        return int(out.value)

    def safety_flags(self):
        buf = (C.c_uint32 * 2)()
        self._check(self._fn("safety_flags")(self._h, buf), "safety_flags")
        return int(buf[0]), int(buf[1])

    def safety_summary(self):
        """Groups per safety-flag bit, flagged groups, and the share of the run's rounds and recorded entries that
        fell in flagged groups (RG_S_*), as a dict."""
        buf = (C.c_uint64 * N_SUMMARY)()
        self._check(self._fn("safety_summary")(self._h, buf), "safety_summary")
        return {n: int(buf[i]) for i, n in enumerate(SUMMARY)}

    def safety_check(self):
        self._check(self._fn("safety_check")(self._h), "safety_check")
        return self.safety_flags()

    def state_hash(self):
        out = np.empty(self.G, dtype=np.uint64)
        self._check(self._fn("state_hash")(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), self.G), "state_hash")
        return out

    def step(self, n_ticks=1):
        self._check(self._fn("step")(self._h, int(n_ticks)), "step")

    def start(self, g, p, cmd):
        i, t, l = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._fn("start")(self._h, g, p, C.c_uint64(cmd), C.byref(i), C.byref(t), C.byref(l)), "start")
        return i.value, t.value, bool(l.value)

    def get_state(self, g, p):
        t, l = C.c_int32(), C.c_int32()
        self._check(self._fn("get_state")(self._h, g, p, C.byref(t), C.byref(l)), "get_state")
        return t.value, bool(l.value)

    def set_connected(self, g, p, on):
        self._check(self._fn("set_connected")(self._h, g, p, 1 if on else 0), "set_connected")

    def set_side(self, g, p, side):
        self._check(self._fn("set_side")(self._h, g, p, side), "set_side")

    def crash(self, g, p):
        self._check(self._fn("crash")(self._h, g, p), "crash")

    def kill(self, g, p):
        """rf.Kill() (src/raft/raft.go:377-380): the dead flag alone; the peer stays on the net."""
        self._check(self._fn("kill")(self._h, g, p), "kill")

    def restart(self, g, p):
        self._check(self._fn("restart")(self._h, g, p), "restart")

    def set_unreliable(self, req_drop=100, reply_drop=100, reply_delay=0):
        self._check(self._fn("set_unreliable")(self._h, req_drop, reply_drop, reply_delay), "set_unreliable")

    def set_unreliable_group(self, g, req_drop=100, reply_drop=100, reply_delay=0):
        self._check(self._fn("set_unreliable_group")(self._h, g, req_drop, reply_drop, reply_delay), "set_unreliable_group")

    def reset_groups(self, groups):
        """Fresh peers, tester record and apply channels in the listed groups, booted at the current tick."""
        a = (C.c_int32 * len(groups))(*[int(g) for g in groups])
        self._check(self._fn("reset_groups")(self._h, a, len(groups)), "reset_groups")

    def set_schedule(self, ops):
        """ops: [(tick, op, a, b), ...] sorted by tick: the tester's verbs for every group at once (rg_set_schedule)."""
        arr = (_abi.SchedOp * max(1, len(ops)))()
        for i, (t, op, a, b) in enumerate(ops):
            arr[i].tick, arr[i].op, arr[i].a, arr[i].b = int(t), int(op), int(a), int(b)
        self._check(self._fn("set_schedule")(self._h, arr, len(ops)), "set_schedule")

    def snapshot_export(self, g, p):
        n = C.c_int64()
        self._check(self._fn("snapshot_export")(self._h, g, p, None, 0, C.byref(n)), "snapshot_export")
        buf = C.create_string_buffer(max(1, n.value))
        self._check(self._fn("snapshot_export")(self._h, g, p, buf, n.value, C.byref(n)), "snapshot_export")
        return buf.raw[:n.value]

    def snapshot_import(self, g, p, blob):
        b = C.create_string_buffer(bytes(blob), max(1, len(blob)))
        self._check(self._fn("snapshot_import")(self._h, g, p, b, len(blob)), "snapshot_import")


class Engine(EngineBase):
    """The HIP engine.  Fails loudly when the library or a gfx950 device is missing."""

    def __init__(self, cfg=None, **kw):
        self._lib = _abi.load_library()
        self.cfg = cfg if cfg is not None else _abi.make_config(**kw)
        self.G, self.P, self.L = self.cfg.n_groups, self.cfg.n_peers, self.cfg.log_cap
        h = C.c_void_p()
        st = self._lib.rg_create(C.byref(self.cfg), C.byref(h))
        if st != 0:
            raise RaftGpuError(st, "rg_create (%s)" % self._lib.rg_strerror(st).decode())
        self._h = h

    def _fn(self, name):
        return getattr(self._lib, "rg_" + name)

    def _rw(self, name, field, ptr, nbytes):
        return self._fn(name)(self._h, field, ptr, nbytes)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rg_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        self._check(self._lib.rg_reset(self._h), "reset")

    def specialize(self):
        """Compile the tick kernel for this engine's configuration (rg_specialize): same results, less time on large engines."""
        self._check(self._lib.rg_specialize(self._h), "specialize")

    def step_noalloc(self, n_ticks=1):
        """rg_step through the entry point that allocates nothing (fixed device scratch)."""
        self._check(self._lib.rg_step_noalloc(self._h, int(n_ticks)), "step_noalloc")

    def sync(self):
        self._check(self._lib.rg_sync(self._h), "sync")

    def now(self):
        t = C.c_int64()
        self._check(self._lib.rg_now(self._h, C.byref(t)), "now")
        return t.value

    def set_stream(self, stream_ptr):
        self._check(self._lib.rg_set_stream(self._h, C.c_void_p(stream_ptr)), "set_stream")

    def step_time_ms(self):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.rg_step_time_ms(self._h, C.byref(ms), C.byref(n)), "step_time_ms")
        return ms.value, n.value

    def start_batch(self, groups, peers, cmds):
        n = len(groups)
        reqs = (StartReq * n)()
        for i in range(n):
            reqs[i].group, reqs[i].peer, reqs[i].command = int(groups[i]), int(peers[i]), int(cmds[i])
        self._check(self._lib.rg_start_batch(self._h, reqs, n), "start_batch")
        return [(r.index, r.term, bool(r.is_leader)) for r in reqs]

    def start_batch_arrays(self, groups, peers, cmds):
        """rg_start_batch for large batches: numpy arrays in, (index, term, is_leader) arrays out, no Python loop."""
        reqs = np.zeros(len(groups), dtype=np.dtype([("group", "<i4"), ("peer", "<i4"), ("command", "<u8"), ("index", "<i4"),
                                                      ("term", "<i4"), ("is_leader", "<i4"), ("pad", "<i4")]))
        reqs["group"], reqs["peer"], reqs["command"] = groups, peers, cmds
        self._check(self._lib.rg_start_batch(self._h, reqs.ctypes.data_as(C.POINTER(StartReq)), len(reqs)), "start_batch")
        return reqs["index"].copy(), reqs["term"].copy(), reqs["is_leader"].astype(bool)

    def drain_applied(self, g, p, cap=None):
        """applyCh: the ApplyMsgs the slot emitted since the last drain, as emitted (needs apply_ring_cap > 0).
        On an overrun of the ring raises RaftGpuError(ERR_OVERRUN) whose `.partial` holds what was still there."""
        cap = cap or max(self.L, self.cfg.apply_ring_cap)
        buf = (ApplyMsgC * cap)()
        n = C.c_int64()
        st = self._lib.rg_drain_applied(self._h, g, p, buf, cap, C.byref(n))
        out = [(int(buf[i].command_index), int(buf[i].command)) for i in range(n.value)]
        if st != 0:
            err = RaftGpuError(st, "drain_applied")
            err.partial = out
            raise err
        return out

    def request_vote(self, g, p, term, candidate, last_log_index, last_log_term):
        """One RequestVote RPC (reference: src/raft/raft.go:200-232)."""
        return self.request_vote_batch([g], [p], [(term, candidate, last_log_index, last_log_term)])[0]

    def request_vote_batch(self, groups, peers, args):
        n = len(groups)
        ga, pa = (C.c_int32 * n)(*groups), (C.c_int32 * n)(*peers)
        aa, ra = (VoteArgs * n)(), (VoteReply * n)()
        for i, a in enumerate(args):
            aa[i].term, aa[i].candidate, aa[i].last_log_index, aa[i].last_log_term = a
        self._check(self._lib.rg_rpc_request_vote(self._h, n, ga, pa, aa, ra), "rpc_request_vote")
        return [(r.term, bool(r.granted)) for r in ra]

    def append_entries(self, g, p, term, leader, prev_log_index, prev_log_term, entries, leader_commit):
        """One AppendEntries RPC (reference: src/raft/raft.go:270-321); entries = [(term, cmd), ...]."""
        return self.append_entries_batch([g], [p], [(term, leader, prev_log_index, prev_log_term, entries, leader_commit)])[0]

    def append_entries_batch(self, groups, peers, reqs):
        n = len(groups)
        ga, pa = (C.c_int32 * n)(*groups), (C.c_int32 * n)(*peers)
        aa, ra = (AppendArgs * n)(), (AppendReply * n)()
        terms, cmds = [], []
        for i, (term, leader, prev, prevt, entries, lc) in enumerate(reqs):
            aa[i].term, aa[i].leader, aa[i].prev_log_index, aa[i].prev_log_term = term, leader, prev, prevt
            aa[i].leader_commit, aa[i].n_entries, aa[i].entry_off = lc, len(entries), len(terms)
            for t, c in entries:
                terms.append(t)
                cmds.append(c)
        ne = len(terms)
        ta, ca = (C.c_int32 * max(1, ne))(*terms), (C.c_uint64 * max(1, ne))(*cmds)
        self._check(self._lib.rg_rpc_append_entries(self._h, n, ga, pa, aa, ta, ca, ne, ra), "rpc_append_entries")
        return [(r.term, bool(r.success), r.conflict_term, r.conflict_first_index) for r in ra]

    # ---- the cross-GPU exchange (RCCL behind the C ABI) ----
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId through the library: 128 bytes for rank 0 to hand to the other ranks."""
        lib = _abi.load_library()
        buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
        st = lib.rg_comm_unique_id(buf)
        if st != 0:
            raise RaftGpuError(st, "rg_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, n_ranks, unique_id=None):
        buf = C.create_string_buffer(bytes(unique_id), _abi.COMM_ID_BYTES) if unique_id is not None else None
        self._check(self._lib.rg_comm_init(self._h, int(rank), int(n_ranks), buf), "comm_init")

    def comm_destroy(self):
        self._check(self._lib.rg_comm_destroy(self._h), "comm_destroy")

    def allreduce_summary(self, nccl_comm=None):
        """(counters, summary) summed over every rank of the communicator: ONE ncclAllReduce on the engine's stream."""
        c, s = (C.c_uint64 * N_COUNTERS)(), (C.c_uint64 * N_SUMMARY)()
        self._check(self._lib.rg_allreduce_summary(self._h, C.c_void_p(nccl_comm), c, s), "allreduce_summary")
        return ({n: int(c[i]) for i, n in enumerate(COUNTERS)}, {n: int(s[i]) for i, n in enumerate(SUMMARY)})

    def device_buffers(self):
        a, b = C.c_void_p(), C.c_void_p()
        self._check(self._lib.rg_device_buffers(self._h, C.byref(a), C.byref(b)), "device_buffers")
        return a.value, b.value
