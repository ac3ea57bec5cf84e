"""ctypes view of include/raftgpu.h: structs, constants and the library loader.

The engine library is `raft_amd/lib/libraftgpu.so`, built in-tree by
`__graft_entry__.build()` (hipcc, gfx950).  There is no CPU fallback: a
missing library or a machine without a usable gfx950 device raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAFTGPU_LIB") or os.path.join(HERE, "lib", "libraftgpu.so")   # env override: A/B builds

# ---- constants mirrored from include/raftgpu.h ---------------------------
TICK_MS = 10
MAX_PEERS = 8
OK, ERR_ARG, ERR_HIP, ERR_NOMEM, ERR_RANGE, ERR_STATE, ERR_SCRATCH, ERR_OVERRUN = 0, -1, -2, -3, -4, -5, -6, -7
FOLLOWER, CANDIDATE, LEADER = 0, 1, 2

FLAG_ELECTION_SAFETY = 0x01
FLAG_LOG_MATCHING = 0x02
FLAG_LEADER_COMPLETENESS = 0x04
FLAG_STATE_MACHINE = 0x08
FLAG_APPLY_ORDER = 0x10
FLAG_REF_PANIC = 0x20
FLAG_LOG_OVERFLOW = 0x40

COUNTERS = [
    "group_ticks", "rpc_sent", "rpc_delivered", "rpc_replied", "rv_rounds",
    "elections_won", "ae_rounds", "entries_started", "entries_replicated",
    "entries_committed", "entries_applied", "persists", "persist_entries",
    "timeouts", "ae_conflicts", "stepdowns", "ae_delivered",
    "rpc_fixed_bytes", "ae_entries_sent",
]
N_COUNTERS = len(COUNTERS)
N_FLAG_BITS = 7
FLAG_NAMES = ["election_safety", "log_matching", "leader_completeness", "state_machine", "apply_order", "ref_panic", "log_overflow"]
# rg_safety_summary / rg_allreduce_summary words (RG_S_*)
SUMMARY = ["groups_" + n for n in FLAG_NAMES] + ["flagged_groups", "rounds_flagged", "rounds_all", "recorded_flagged", "recorded_all"]
N_SUMMARY = len(SUMMARY)
COMM_ID_BYTES = 128

# field id -> (name, kind); kinds: p32 [G][P] i32, p64 [G][P] u64, m32 [G][P][P],
# l32 [G][P][L] i32, l64 [G][P][L] u64, g32 [G] i32, r64 [G][L] u64
_PEER_I32 = ["term", "vote", "state", "commit", "applied", "loglen", "timer",
             "timeout", "hbtimer", "stop", "cnt", "dead", "restart", "first",
             "monalive", "elnext", "monnext", "rngctr", "roundseq", "snaphas",
             "snapterm", "snapvote", "snaplen", "connected", "side"]
FIELDS = {name: (i, "p32") for i, name in enumerate(_PEER_I32)}
FIELDS.update({
    "applyhash": (32, "p64"),
    "nextindex": (40, "m32"), "matchindex": (41, "m32"),
    "logterm": (48, "l32"), "snaplogterm": (49, "l32"),
    "logcmd": (56, "l64"), "snaplogcmd": (57, "l64"),
    "startseq": (64, "g32"), "partnext": (65, "g32"), "partepoch": (66, "g32"),
    "recmax": (67, "g32"), "gflags": (68, "g32"), "schedpos": (69, "g32"), "schedregs": (70, "g32"),
    "applyrec": (72, "r64"),
})
START_ONE, START_ALL = 0, 1
LOG_AUTO, LOG_ROWS, LOG_COLUMNS = 0, 1, 2

# scenario schedule ops and selectors
(OP_CONNECT, OP_DISCONNECT, OP_CRASH, OP_RESTART, OP_START, OP_START_ANY, OP_START_ALL, OP_PICK_LEADER, OP_PICK_RANDOM,
 OP_PICK_OTHER, OP_RESTART_IF_FEWER, OP_CONNECT_IF_FEWER, OP_KILL) = range(1, 14)
SEL_ALL = 0x80


def SEL_REG(r, k=0):
    return 0x100 | (r << 4) | k


class Config(C.Structure):
    _fields_ = [
        ("n_groups", C.c_int32), ("n_peers", C.c_int32), ("log_cap", C.c_int32), ("device", C.c_int32),
        ("group_base", C.c_int64), ("seed", C.c_uint64),
        ("req_drop_permille", C.c_int32), ("reply_drop_permille", C.c_int32), ("reply_delay_permille", C.c_int32),
        ("start_period", C.c_int32), ("start_first_tick", C.c_int32), ("start_max", C.c_int32), ("start_mode", C.c_int32),
        ("part_period", C.c_int32), ("part_jitter", C.c_int32),
        ("churn_period", C.c_int32), ("churn_disconnect_permille", C.c_int32),
        ("churn_restart_permille", C.c_int32), ("churn_crash_permille", C.c_int32),
        ("account_bytes", C.c_int32), ("apply_ring_cap", C.c_int32), ("log_layout", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class VoteArgs(C.Structure):
    _fields_ = [("term", C.c_int32), ("candidate", C.c_int32), ("last_log_index", C.c_int32), ("last_log_term", C.c_int32)]


class VoteReply(C.Structure):
    _fields_ = [("term", C.c_int32), ("granted", C.c_int32)]


class AppendArgs(C.Structure):
    _fields_ = [("term", C.c_int32), ("leader", C.c_int32), ("prev_log_index", C.c_int32), ("prev_log_term", C.c_int32),
                ("leader_commit", C.c_int32), ("n_entries", C.c_int32), ("entry_off", C.c_int64)]


class AppendReply(C.Structure):
    _fields_ = [("term", C.c_int32), ("success", C.c_int32), ("conflict_term", C.c_int32), ("conflict_first_index", C.c_int32)]


class ApplyMsgC(C.Structure):
    _fields_ = [("command_index", C.c_int64), ("command", C.c_uint64)]


class SchedOp(C.Structure):
    _fields_ = [("tick", C.c_int32), ("op", C.c_int32), ("a", C.c_int32), ("b", C.c_int32)]


class StartReq(C.Structure):
    _fields_ = [("group", C.c_int32), ("peer", C.c_int32), ("command", C.c_uint64),
                ("index", C.c_int32), ("term", C.c_int32), ("is_leader", C.c_int32), ("pad", C.c_int32)]


def make_config(**kw):
    cfg = Config()
    cfg.n_groups, cfg.n_peers, cfg.log_cap = 1, 3, 64
    cfg.seed = 1
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError("unknown config field %r" % k)
        setattr(cfg, k, v)
    return cfg


# every symbol include/raftgpu.h declares
EXPORTS = [
    "rg_device_count", "rg_version", "rg_strerror", "rg_create", "rg_destroy", "rg_reset", "rg_set_stream",
    "rg_step", "rg_step_noalloc", "rg_specialize", "rg_sync", "rg_now", "rg_step_time_ms", "rg_start", "rg_start_batch", "rg_get_state",
    "rg_set_connected", "rg_set_side", "rg_crash", "rg_restart", "rg_kill", "rg_set_unreliable", "rg_set_unreliable_group", "rg_reset_groups", "rg_set_schedule", "rg_drain_applied",
    "rg_snapshot_export", "rg_snapshot_import", "rg_rpc_request_vote", "rg_rpc_append_entries",
    "rg_read_state", "rg_write_state", "rg_state_hash", "rg_counters", "rg_rpc_bytes", "rg_safety_check", "rg_safety_flags",
    "rg_device_buffers", "rg_safety_summary", "rg_comm_unique_id", "rg_comm_init", "rg_comm_destroy", "rg_allreduce_summary",
]

_lib = None


def load_library(path=None):
    """dlopen the HIP engine.  Raises if it is not built: nothing substitutes for it."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("HIP engine library missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % p)
    lib = C.CDLL(p)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    P = C.POINTER
    sig = {
        "rg_device_count": ([], C.c_int),
        "rg_version": ([], C.c_char_p),
        "rg_strerror": ([C.c_int], C.c_char_p),
        "rg_create": ([P(Config), P(vp)], C.c_int),
        "rg_destroy": ([vp], C.c_int),
        "rg_reset": ([vp], C.c_int),
        "rg_set_stream": ([vp, vp], C.c_int),
        "rg_step": ([vp, i32], C.c_int),
        "rg_step_noalloc": ([vp, i32], C.c_int),
        "rg_specialize": ([vp], C.c_int),
        "rg_sync": ([vp], C.c_int),
        "rg_now": ([vp, P(i64)], C.c_int),
        "rg_step_time_ms": ([vp, P(C.c_double), P(i64)], C.c_int),
        "rg_start": ([vp, i32, i32, C.c_uint64, P(i32), P(i32), P(i32)], C.c_int),
        "rg_start_batch": ([vp, P(StartReq), i64], C.c_int),
        "rg_get_state": ([vp, i32, i32, P(i32), P(i32)], C.c_int),
        "rg_set_connected": ([vp, i32, i32, i32], C.c_int),
        "rg_set_side": ([vp, i32, i32, i32], C.c_int),
        "rg_crash": ([vp, i32, i32], C.c_int),
        "rg_restart": ([vp, i32, i32], C.c_int),
        "rg_kill": ([vp, i32, i32], C.c_int),
        "rg_set_unreliable": ([vp, i32, i32, i32], C.c_int),
        "rg_set_unreliable_group": ([vp, i32, i32, i32, i32], C.c_int),
        "rg_reset_groups": ([vp, P(i32), i64], C.c_int),
        "rg_set_schedule": ([vp, P(SchedOp), i64], C.c_int),
        "rg_drain_applied": ([vp, i32, i32, P(ApplyMsgC), i64, P(i64)], C.c_int),
        "rg_snapshot_export": ([vp, i32, i32, vp, i64, P(i64)], C.c_int),
        "rg_snapshot_import": ([vp, i32, i32, vp, i64], C.c_int),
        "rg_rpc_request_vote": ([vp, i64, P(i32), P(i32), P(VoteArgs), P(VoteReply)], C.c_int),
        "rg_rpc_append_entries": ([vp, i64, P(i32), P(i32), P(AppendArgs), P(i32), P(C.c_uint64), i64, P(AppendReply)], C.c_int),
        "rg_read_state": ([vp, i32, vp, i64], C.c_int),
        "rg_write_state": ([vp, i32, vp, i64], C.c_int),
        "rg_state_hash": ([vp, P(C.c_uint64), i64], C.c_int),
        "rg_counters": ([vp, P(C.c_uint64)], C.c_int),
        "rg_rpc_bytes": ([vp, i64, P(C.c_uint64)], C.c_int),
        "rg_safety_check": ([vp], C.c_int),
        "rg_safety_flags": ([vp, P(C.c_uint32)], C.c_int),
        "rg_device_buffers": ([vp, P(vp), P(vp)], C.c_int),
        "rg_safety_summary": ([vp, P(C.c_uint64)], C.c_int),
        "rg_comm_unique_id": ([vp], C.c_int),
        "rg_comm_init": ([vp, i32, i32, vp], C.c_int),
        "rg_comm_destroy": ([vp], C.c_int),
        "rg_allreduce_summary": ([vp, vp, P(C.c_uint64), P(C.c_uint64)], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    if path is None:
        _lib = lib
    return lib
