/*
 * raftgpu.h -- C ABI of the batched lockstep Raft engine for MI355X (gfx950).
 *
 * One engine advances n_groups independent Raft replica groups of n_peers
 * peers each in discrete 10 ms ticks.  Every (group, peer) pair is a drop-in
 * for one instance of the reference peer (reference: src/raft/raft.go:57-86)
 * and is driven through the reference's own verbs:
 *
 *   Make        src/raft/raft.go:700   -> rg_create / rg_restart
 *   Start       src/raft/raft.go:344   -> rg_start / rg_start_batch
 *   GetState    src/raft/raft.go:90    -> rg_get_state
 *   Kill        src/raft/raft.go:377   -> rg_kill (the tester's crash1, config.go:99-127 -> rg_crash)
 *   ApplyMsg    src/raft/raft.go:42    -> rg_drain_applied
 *   persist / readPersist  src/raft/raft.go:107,129 and the Persister of
 *               src/raft/persister.go:20-70 -> rg_snapshot_export/import
 *   RequestVote / AppendEntries handlers src/raft/raft.go:200,270
 *                                      -> rg_rpc_request_vote / rg_rpc_append_entries
 *   harness verbs connect/disconnect/crash1/start1 src/raft/config.go:99-278
 *                                      -> rg_set_connected / rg_crash / rg_restart
 *
 * All entry points are extern "C", take plain pointers and sizes, return an
 * int status (0 = ok, <0 = error) and never abort.  Buffers are allocated by
 * the caller.  One mutex inside the engine serialises calls, so concurrent
 * callers (goroutines behind cgo) are safe.  There is no CPU fallback: every
 * call fails with RG_ERR_HIP when no gfx950 device is usable.
 */
#ifndef RAFTGPU_H
#define RAFTGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- lockstep time base (all reference timing literals are multiples of
 * these; src/raft/raft.go:421,524,532-534,567,665,667,675) ---------------- */
#define RG_TICK_MS            10   /* one tick of simulated time               */
#define RG_ELECTION_PERIOD_MS 50   /* election loop sleep         (:667)        */
#define RG_ELECTION_LEADER_EXTRA_MS 10 /* extra sleep when it saw state==2 (:665) */
#define RG_MONITOR_POLL_MS    20   /* leader loop idle sleep      (:524)        */
#define RG_HB_IDLE_MS        100   /* heartbeat gate, idle        (:421)        */
#define RG_HB_BACKLOG_MS      50   /* heartbeat gate, uncommitted (:421)        */
#define RG_CANDIDATE_WINDOW_MS 1000/* timeout while candidate     (:567)        */
#define RG_TIMEOUT_BASE_MS   500   /* nextRandom(): Intn(600)+500 (:532-534)    */
#define RG_TIMEOUT_SPAN_MS   600
#define RG_RPC_WAIT_MS        25   /* longest a sender waits for one Call (:473,:608); a Call the net fails earlier costs less (labrpc.go:226-229,273-275) */
#define RG_NET_DELAY_SPAN_MS  27   /* unreliable net: each Call sleeps 0..26 ms (labrpc.go:222)     */
#define RG_MAX_PEERS           8

/* ---- status codes ------------------------------------------------------ */
#define RG_OK            0
#define RG_ERR_ARG      -1   /* null pointer / bad size / bad config          */
#define RG_ERR_HIP      -2   /* no device, launch or runtime failure          */
#define RG_ERR_NOMEM    -3
#define RG_ERR_RANGE    -4   /* group / peer / index out of range             */
#define RG_ERR_STATE    -5   /* verb not legal in the peer's current state    */
#define RG_ERR_OVERRUN  -7   /* rg_drain_applied: ApplyMsgs were overwritten in the ring before they were drained */
#define RG_ERR_SCRATCH  -6   /* rg_step_noalloc: the engine needs more than the fixed device scratch */

/* ---- peer roles (Raft.state, src/raft/raft.go:82) ----------------------- */
#define RG_FOLLOWER  0
#define RG_CANDIDATE 1
#define RG_LEADER    2

/* ---- safety / diagnostic flag bits, one word per group ------------------ */
#define RG_FLAG_ELECTION_SAFETY     0x01u /* two connected leaders in one term (src/raft/config.go:317-320) */
#define RG_FLAG_LOG_MATCHING        0x02u /* same index+term but different command or different predecessor */
#define RG_FLAG_LEADER_COMPLETENESS 0x04u /* a live leader lacks an entry some peer already applied         */
#define RG_FLAG_STATE_MACHINE       0x08u /* two peers applied different commands at one index (src/raft/config.go:177-182) */
#define RG_FLAG_APPLY_ORDER         0x10u /* apply skipped an index (src/raft/config.go:184,191-193)        */
#define RG_FLAG_REF_PANIC           0x20u /* state in which the reference would index past len(logs)        */
#define RG_FLAG_LOG_OVERFLOW        0x40u /* a Start was refused because the fixed log capacity is full     */
#define RG_N_FLAG_BITS 7

/* ---- counters (uint64 each) --------------------------------------------- */
enum {
  RG_C_GROUP_TICKS = 0,     /* groups x ticks advanced                                  */
  RG_C_RPC_SENT,            /* every Call made, lost ones included (labrpc.go:153)       */
  RG_C_RPC_DELIVERED,       /* handler ran on the target (labrpc.go:406)                 */
  RG_C_RPC_REPLIED,         /* complete round trips                                      */
  RG_C_RV_ROUNDS,           /* candidate rounds started (raft.go:565)                    */
  RG_C_ELECTIONS_WON,       /* raft.go:641-658                                           */
  RG_C_AE_ROUNDS,           /* leader replication rounds (raft.go:421-512)               */
  RG_C_ENTRIES_STARTED,     /* accepted Start calls (raft.go:351-358)                    */
  RG_C_ENTRIES_REPLICATED,  /* entries carried by successful AppendEntries               */
  RG_C_ENTRIES_COMMITTED,   /* leader commitIndex advance (raft.go:409-411)              */
  RG_C_ENTRIES_APPLIED,     /* ApplyMsg emitted (raft.go:677-685)                        */
  RG_C_PERSISTS,            /* persist() calls (raft.go:107)                             */
  RG_C_PERSIST_ENTRIES,     /* log entries newly copied into the snapshot arrays         */
  RG_C_TIMEOUTS,            /* follower -> candidate transitions (raft.go:557-561)       */
  RG_C_AE_CONFLICTS,        /* AppendEntries answered with a conflict hint (raft.go:303) */
  RG_C_STEPDOWNS,           /* sender saw a higher term in a reply (raft.go:482,622)     */
  RG_C_AE_DELIVERED,        /* AppendEntries handler runs (subset of RPC_DELIVERED)      */
  RG_C_RPC_FIXED_BYTES,     /* fixed-width bytes on the wire: arguments of every Call + every reply the net carried, late ones included (labrpc.go:153-154,286-290) */
  RG_C_AE_ENTRIES_SENT,     /* log entries carried by AppendEntries Calls                */
  RG_N_COUNTERS
};

/* ---- safety summary (uint64 each): what rg_safety_summary and rg_allreduce_summary report ------------------ */
enum {
  RG_S_BIT_GROUPS = 0,        /* [RG_N_FLAG_BITS] groups whose flag word has bit b set                          */
  RG_S_FLAGGED_GROUPS = RG_N_FLAG_BITS, /* groups with any flag                                                  */
  RG_S_ROUNDS_FLAGGED,        /* RequestVote + AppendEntries rounds sent by peers of flagged groups ...          */
  RG_S_ROUNDS_ALL,            /* ... and of all groups (a round is up to n_peers - 1 Calls)                      */
  RG_S_RECORDED_FLAGGED,      /* highest index the tester recorded as applied (config.go:170-194), summed over flagged groups ... */
  RG_S_RECORDED_ALL,          /* ... and over all groups                                                         */
  RG_N_SUMMARY
};

/* ---- state fields for rg_read_state / rg_write_state -------------------- *
 * Host layout: per-peer fields int32 [n_groups][n_peers]; per-group fields
 * int32 [n_groups]; matrices int32 [n_groups][n_peers(owner)][n_peers];
 * logs [n_groups][n_peers][log_cap] (int32 terms, uint64 commands).
 * TIMER, HBTIMER, ELNEXT, MONNEXT, TIMEOUT are in ms of simulated time
 * (tick x 10), PARTNEXT in ticks.  DEAD: 0 alive, 1 crashed (crash1: its server is
 * gone from the net), 2 killed (Kill(): dead flag only, Calls to it still return). */
enum {
  RG_F_TERM = 0, RG_F_VOTE, RG_F_STATE, RG_F_COMMIT, RG_F_APPLIED, RG_F_LOGLEN,
  RG_F_TIMER, RG_F_TIMEOUT, RG_F_HBTIMER, RG_F_STOP, RG_F_CNT, RG_F_DEAD,
  RG_F_RESTART, RG_F_FIRST, RG_F_MONALIVE, RG_F_ELNEXT, RG_F_MONNEXT,
  RG_F_RNGCTR, RG_F_ROUNDSEQ, RG_F_SNAPHAS, RG_F_SNAPTERM, RG_F_SNAPVOTE,
  RG_F_SNAPLEN, RG_F_CONNECTED, RG_F_SIDE,
  RG_F_PEER_I32_END,                 /* fields above: int32 [G][P]           */
  RG_F_APPLYHASH = 32,               /* uint64 [G][P]                        */
  RG_F_NEXTINDEX = 40, RG_F_MATCHINDEX, /* int32 [G][P][P]                   */
  RG_F_LOGTERM = 48, RG_F_SNAPLOGTERM,  /* int32 [G][P][L]                   */
  RG_F_LOGCMD = 56, RG_F_SNAPLOGCMD,    /* uint64 [G][P][L]                  */
  RG_F_STARTSEQ = 64, RG_F_PARTNEXT, RG_F_PARTEPOCH, RG_F_RECMAX, RG_F_GFLAGS, /* int32 [G] */
  RG_F_SCHEDPOS, RG_F_SCHEDREGS,     /* int32 [G]: next op of the schedule, the group's four peer registers (4 bits each, 15 = none) */
  RG_F_APPLYREC = 72,                /* uint64 [G][L]                        */
  /* read-only, int32 [G][P]: the engine's cached copies of log words (term of the last entry, first index from
   * which every entry carries currentTerm, term of the entry at commitIndex), for consistency checks */
  RG_F_CACHE_LASTTERM = 80, RG_F_CACHE_FIRSTOWN, RG_F_CACHE_COMMITTERM
};

/* ---- Start injection modes (device-side client workload) ---------------- */
#define RG_START_ONE  0  /* first connected peer, in index order, that accepts (config.go:436-451) */
#define RG_START_ALL  1  /* every live peer that claims leadership (test_test.go:697-703)          */

/* ---- log layouts (rg_config.log_layout) ---------------------------------- */
#define RG_LOG_AUTO    0
#define RG_LOG_ROWS    1  /* entry i of all peers of a group in one row: replication and apply of index i touch one line */
#define RG_LOG_COLUMNS 2  /* per-peer contiguous term[] and command[] arrays: replays, catch-up copies and conflict walks stream */

/* Engine configuration: one plain struct across the ABI. */
typedef struct rg_config {
  int32_t  n_groups;            /* groups owned by this engine (this GPU's shard)                 */
  int32_t  n_peers;             /* 2..8 peers per group                                           */
  int32_t  log_cap;             /* fixed log capacity per peer, entry 0 is the sentinel           */
  int32_t  device;              /* HIP device ordinal                                             */
  int64_t  group_base;          /* global id of local group 0; all seeded streams key on global id */
  uint64_t seed;
  /* network model (labrpc.go:216-307) */
  int32_t  req_drop_permille;   /* request lost before the handler runs   (:226)                  */
  int32_t  reply_drop_permille; /* reply lost after the handler ran       (:273)                  */
  int32_t  reply_delay_permille;/* long reordering: reply later than the 25 ms wait (:276)        */
  /* client workload generated on the device */
  int32_t  start_period;        /* ticks between Start injections, 0 = none                       */
  int32_t  start_first_tick;
  int32_t  start_max;           /* stop after this many accepted Starts per group                 */
  int32_t  start_mode;          /* RG_START_ONE / RG_START_ALL                                    */
  /* rolling partitions (kvraft/test_test.go:124-143): every part_period +
   * hash % part_jitter ticks each peer is put on side hash & 1             */
  int32_t  part_period;         /* 0 = none                                                       */
  int32_t  part_jitter;
  /* churn (test_test.go:893-919): every churn_period ticks, per group      */
  int32_t  churn_period;        /* 0 = none                                                       */
  int32_t  churn_disconnect_permille;
  int32_t  churn_restart_permille;
  int32_t  churn_crash_permille;
  /* wire-byte accounting for rg_rpc_bytes (RG_C_RPC_FIXED_BYTES, RG_C_AE_ENTRIES_SENT): off by default */
  int32_t  account_bytes;
  /* applyCh for host readers (rg_drain_applied): ApplyMsgs each peer may have pending between two drains, 0 = the
   * engine keeps no ring (device-generated workloads, whose tester side runs on the device).  16 bytes per
   * message, n_groups x n_peers x apply_ring_cap messages of device memory. */
  int32_t  apply_ring_cap;
  /* how the logs lie in device memory (no effect on any result): RG_LOG_AUTO picks columns for logs of 2,048 entries
   * or more and for engines with crash / restart churn (restarted peers replay their whole log, raft.go:717 with
   * :672-688), rows otherwise */
  int32_t  log_layout;
} rg_config;

typedef struct rg_engine rg_engine;

/* RequestVoteArgs / Reply (src/raft/raft.go:161-177), fixed-width. */
typedef struct rg_vote_args  { int32_t term, candidate, last_log_index, last_log_term; } rg_vote_args;
typedef struct rg_vote_reply { int32_t term, granted; } rg_vote_reply;
/* AppendEntriesArgs / Reply (src/raft/raft.go:179-195).  Entries of request i
 * are entry_terms/entry_cmds[entry_off .. entry_off + n_entries).           */
typedef struct rg_append_args {
  int32_t term, leader, prev_log_index, prev_log_term, leader_commit, n_entries;
  int64_t entry_off;
} rg_append_args;
typedef struct rg_append_reply { int32_t term, success, conflict_term, conflict_first_index; } rg_append_reply;
/* ApplyMsg (src/raft/raft.go:42-46); CommandValid is always true here. */
typedef struct rg_apply_msg { int64_t command_index; uint64_t command; } rg_apply_msg;
/* One host command for rg_start_batch. */
typedef struct rg_start_req { int32_t group, peer; uint64_t command; int32_t index, term, is_leader, pad; } rg_start_req;

/* Library / device probe: number of usable gfx950 devices, or RG_ERR_HIP. */
int rg_device_count(void);
const char *rg_version(void);
const char *rg_strerror(int status);

/* Make (raft.go:700-731) for every peer of every group, all connected, tick 0. */
int rg_create(const rg_config *cfg, rg_engine **out);
int rg_destroy(rg_engine *e);
/* Back to the state rg_create leaves (tick 0, empty persisters), same config.  Log entries beyond a log's length and
 * tester records beyond the group's highest recorded index keep what the run before left in them (a Go slice has no
 * such elements; nothing in the engine reads them before they are written again): raw dumps of those arrays
 * (rg_read_state) are meaningful up to the lengths only. */
int rg_reset(rg_engine *e);
/* Launch on a caller-owned HIP stream (hipStream_t), NULL = the engine's own. */
int rg_set_stream(rg_engine *e, void *hip_stream);

/* Advance every group n_ticks ticks (election timers raft.go:536-669, leader
 * loop :387-530, handlers :200-321, apply :672-688, persist :107-123) in ONE
 * kernel launch; state stays resident in HBM between calls.  Asynchronous on
 * the engine's stream; rg_sync waits.
 * A call of at most 12 ticks (the tester's way of driving an instance: one tick at a time, verbs in between) only runs
 * the groups that have something to do inside those ticks -- every group's next such tick is kept next to its state
 * -- and leaves the others alone: groups whose peers are all crashed (slots waiting for a config) cost 4 bytes of
 * reading each.  What the untouched groups would have done meanwhile changes nothing but their own wake-up times (and,
 * with a device client, appends the client's Starts); it is carried out when the group next runs or before the host
 * reads state, so every call returns what tick-by-tick execution returns.  RAFTGPU_DUE_MAX_TICKS in the environment
 * changes the limit (0: every call runs every group). */
int rg_step(rg_engine *e, int32_t n_ticks);
/* rg_step without any allocation.  No counterpart in the reference: it is for callers that cannot allocate
 * device or host memory while they step -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same kernels and same result as rg_step; the work queue and staging area that launches of
 * more groups than the chip holds at once need live in a fixed 8 MiB device array (RG_NOALLOC_SCRATCH_BYTES,
 * one copy per GPU, held until the process ends) instead of the buffers rg_create allocated; the launch is not
 * timed (rg_step_time_ms does not see it).  Engines too large for the array (more than about 4,600 sets of
 * 4 wavefronts' groups, e.g. 220,000 five-peer groups) get RG_ERR_SCRATCH and nothing is launched; engines that
 * fit the chip at once, and calls short enough for a plain launch (fewer than 2,000 ticks), use no scratch and are
 * never refused.  All engines on one GPU share the array: a launch that uses it waits on the GPU (a stream wait on
 * a per-device event, no allocation) until the previous launch that used it -- of any engine, on any stream -- has
 * ended, so overlapping calls are serialised, not raced. */
#define RG_NOALLOC_SCRATCH_BYTES (8u << 20)
int rg_step_noalloc(rg_engine *e, int32_t n_ticks);
/* Compiles the tick kernel for THIS engine's configuration.  No counterpart in the reference: Go compiles
 * raft.go once for every configuration.  What rg_create fixed -- peers, log capacity, the net's rates, the
 * parameters of the client, partition and churn schedules, byte accounting -- becomes a compile-time constant
 * of a kernel built at run time (hiprtc, about a second) from the device sources embedded in the library;
 * rg_step then launches it in place of the generic instantiation.  Same code, same results, 3-9 % less
 * time on large engines (fewer scalar registers, folded arithmetic).  A later change of what was compiled in
 * (rg_set_unreliable, an apply ring, a schedule or per-group rates appearing) rebuilds the kernel at the next
 * rg_step; rg_step_noalloc only uses a kernel that is already built and fits.  RG_ERR_STATE: hiprtc could not be
 * loaded; RG_ERR_HIP: the compile or the module load failed (the compiler's log goes to stderr); RG_ERR_ARG: the
 * environment's RAFTGPU_SPEC_DEFS (extra compiler options, a tuning aid) holds an option that is not on the list
 * of options the library passes on; in all cases the engine keeps running on the generic kernel.  Setting RAFTGPU_SPECIALIZE=1 in the environment does the same
 * for every engine at its first launch. */
int rg_specialize(rg_engine *e);
int rg_sync(rg_engine *e);
int rg_now(rg_engine *e, int64_t *tick);
/* Wall time (ms) of the kernels launched by rg_step so far, from HIP events
 * recorded on the launch stream, and the number of launches.  Synchronises.
 * The two events around a launch cost the stream about 8 us; of the launches of at most 12 ticks (above) one in 16 is
 * timed and counted for the untimed ones before it (RAFTGPU_DUE_TIME_EVERY=1 in the environment: every one). */
int rg_step_time_ms(rg_engine *e, double *total_ms, int64_t *launches);

/* Start (raft.go:344-364): index/term/is_leader as the reference returns them. */
int rg_start(rg_engine *e, int32_t group, int32_t peer, uint64_t command,
             int32_t *index, int32_t *term, int32_t *is_leader);
int rg_start_batch(rg_engine *e, rg_start_req *reqs, int64_t n);
/* GetState (raft.go:90-100). */
int rg_get_state(rg_engine *e, int32_t group, int32_t peer, int32_t *term, int32_t *is_leader);

/* Harness verbs (config.go:235-278, 99-127, 136-215).  rg_crash = crash1:
 * disconnect, delete the server from the net, drop volatile state, keep the
 * snapshot arrays.  rg_restart = start1: crash1, then Make + readPersist from
 * the snapshot arrays; the peer stays disconnected until rg_set_connected, as
 * in the harness. */
int rg_set_connected(rg_engine *e, int32_t group, int32_t peer, int32_t connected);
int rg_set_side(rg_engine *e, int32_t group, int32_t peer, int32_t side);
int rg_crash(rg_engine *e, int32_t group, int32_t peer);
int rg_restart(rg_engine *e, int32_t group, int32_t peer);
/* Kill (raft.go:377-380): the dead flag and nothing else.  The peer's loops end (:395, :548, :674), GetState and
 * Start answer not-leader, its handlers return at once (:203, :273) -- but it stays connected and its server stays
 * registered, so a Call to it returns TRUE with the reply as the caller initialised it: a leader counts the peer
 * active (:476-477), reads Success = false with ConflictTerm 0 and backs nextIndex off to 1 (:493-499); a candidate
 * counts it active without a vote.  No effect on a peer that is already dead.  rg_restart brings it back. */
int rg_kill(rg_engine *e, int32_t group, int32_t peer);
/* Network.Reliable / LongReordering (labrpc.go:169-181), all groups. */
int rg_set_unreliable(rg_engine *e, int32_t req_drop_permille,
                      int32_t reply_drop_permille, int32_t reply_delay_permille);

/* The same for ONE group: every tester config of the reference owns its network (config.go:70, 288-294), so a
 * host that runs one config per group sets them per group.  A group without a call of its own uses the engine's
 * rates; rg_reset_groups and rg_reset drop the group's own. */
int rg_set_unreliable_group(rg_engine *e, int32_t group, int32_t req_drop_permille,
                            int32_t reply_drop_permille, int32_t reply_delay_permille);
/* make_config (config.go:60-96) inside a running engine: the listed groups (no repeats) get fresh peers -- Make with
 * empty Persisters at the current tick, all connected -- a fresh tester record, cleared flags and logs, and fresh
 * apply channels (ApplyMsgs of the old peers that were not drained are dropped).  Other groups are not touched;
 * counters keep counting. */
int rg_reset_groups(rg_engine *e, const int32_t *groups, int64_t n);

/* ---- scenario schedules: the tester's verbs for ALL groups at once ---------------------------------------------
 * A schedule is a list of ops sorted by tick; at the start of tick `tick` (before the rolling partitions, the churn
 * and the client Starts of that tick) every group of the engine executes the op on its own peers, inside the tick
 * kernel: a scenario of test_test.go (connect / disconnect / crash1 / start1 / Start / checkOneLeader,
 * config.go:99-278, 302-332, 430-477) runs on every group in one launch.  Peers are named by a selector: 0..7 an
 * index, RG_SEL_ALL every peer, RG_SEL_REG(r, k) = (register r + k) mod n_peers, where the four registers of a
 * group are set by the PICK ops (an op on an empty register does nothing).  Commands are hash(group, 1, b). */
#define RG_SEL_ALL 0x80
#define RG_SEL_REG(r, k) (0x100 | ((r) << 4) | (k))
enum {
  RG_OP_CONNECT = 1,       /* a: selector                                   config.go:235-256 */
  RG_OP_DISCONNECT,        /* a: selector                                   config.go:258-278 */
  RG_OP_CRASH,             /* a: selector; crash1                            config.go:99-127  */
  RG_OP_RESTART,           /* a: selector; start1 (stays disconnected)       config.go:136-215 */
  RG_OP_START,             /* a: selector, b: command number; rf.Start       raft.go:344-364   */
  RG_OP_START_ANY,         /* b: command number; the first connected peer, in index order, that accepts  config.go:436-451 */
  RG_OP_START_ALL,         /* a: register, b: command number; every live peer is asked, the register gets the last one that accepted  test_test.go:697-703 */
  RG_OP_PICK_LEADER,       /* a: register := the connected leader of the highest term (lowest index), or none  config.go:302-332 */
  RG_OP_PICK_RANDOM,       /* a: register := hash(group, op number, b) mod n_peers */
  RG_OP_PICK_OTHER,        /* a: register := the first connected live peer after selector b, in cyclic order, or none */
  RG_OP_RESTART_IF_FEWER,  /* a: selector, b: n; if that peer is dead and fewer than n peers are alive: start1 + connect  test_test.go:718-725 */
  RG_OP_CONNECT_IF_FEWER,  /* a: selector, b: n; if that peer is disconnected and fewer than n peers are connected: connect  test_test.go:805-811 */
  RG_OP_KILL,              /* a: selector; rf.Kill(): dead flag only, the peer stays on the net          raft.go:377-380   */
  RG_OP_END
};
typedef struct rg_sched_op { int32_t tick, op, a, b; } rg_sched_op;
/* Replaces the engine's schedule (n = 0: none).  Ops with tick < now are never executed.  Engines with a schedule
 * run plain launches. */
int rg_set_schedule(rg_engine *e, const rg_sched_op *ops, int64_t n);

/* applyCh (raft.go:672-688): copies the ApplyMsgs the (group, peer) slot emitted since the last drain into buf,
 * in the order and with the commands of the moment they were emitted (raft.go:680-684): an entry applied and later
 * overwritten in the log is delivered as it was applied, and a peer restarted in between (by rg_restart or by the
 * device-side churn) delivers its replay from index 1 after what its predecessor emitted.  Needs
 * rg_config.apply_ring_cap > 0 (else RG_ERR_STATE).  If more than apply_ring_cap messages were emitted since the
 * last drain the oldest are gone: the call returns RG_ERR_OVERRUN once, with the messages still there. */
int rg_drain_applied(rg_engine *e, int32_t group, int32_t peer,
                     rg_apply_msg *buf, int64_t cap, int64_t *n_out);

/* Persister blob (persister.go:34-45): int32 term, int32 vote, int32 len,
 * int32 0, then len x {int32 term, int32 0, uint64 command}.  Export with
 * buf == NULL returns the size.  Import replaces the peer's snapshot arrays;
 * it takes effect at the next rg_restart, like swapping the Persister. */
int rg_snapshot_export(rg_engine *e, int32_t group, int32_t peer,
                       void *buf, int64_t cap, int64_t *size_out);
int rg_snapshot_import(rg_engine *e, int32_t group, int32_t peer,
                       const void *buf, int64_t size);

/* The reference's exported RPC handlers, batched: request i is handled by
 * peer peers[i] of group groups[i]; requests to one peer are handled in
 * array order.  Dead peers leave the reply untouched (raft.go:203,273). */
int rg_rpc_request_vote(rg_engine *e, int64_t n, const int32_t *groups, const int32_t *peers,
                        const rg_vote_args *args, rg_vote_reply *replies);
int rg_rpc_append_entries(rg_engine *e, int64_t n, const int32_t *groups, const int32_t *peers,
                          const rg_append_args *args, const int32_t *entry_terms,
                          const uint64_t *entry_cmds, int64_t n_entries_total,
                          rg_append_reply *replies);

/* Bulk state access in the host layout given with the RG_F_* ids. */
int rg_read_state(rg_engine *e, int32_t field, void *dst, int64_t dst_bytes);
int rg_write_state(rg_engine *e, int32_t field, const void *src, int64_t src_bytes);
/* One 64-bit digest per group over every Raft-visible word (scalars, the
 * leader matrices, log and snapshot prefixes, apply digests). */
int rg_state_hash(rg_engine *e, uint64_t *out, int64_t n_groups);

int rg_counters(rg_engine *e, uint64_t *out /* RG_N_COUNTERS */);
/* cfg.bytesTotal(), config.go:292-294 = labrpc's GetTotalBytes (labrpc.go:153-154,286-290): bytes of the
 * arguments of every Call made plus bytes of every reply the net carried.  The reference counts gob bytes;
 * here a message costs its fixed-width fields (RG_*_BYTES below) plus, per log entry, 4 bytes of term and
 * `command_bytes` of payload (the size the caller's commands have; the engine carries 8-byte handles). */
#define RG_VOTE_ARGS_BYTES 16
#define RG_VOTE_REPLY_BYTES 8
#define RG_APPEND_ARGS_BYTES 24
#define RG_APPEND_REPLY_BYTES 16
int rg_rpc_bytes(rg_engine *e, int64_t command_bytes, uint64_t *total);   /* RG_ERR_STATE unless rg_config.account_bytes */
/* Full log-matching and leader-completeness scan; ORs into the group flags. */
int rg_safety_check(rg_engine *e);
/* out[0] = OR of all group flag words, out[1] = number of flagged groups. */
int rg_safety_flags(rg_engine *e, uint32_t *out /* 2 */);
/* Flagged groups per flag bit and the share of the run's work they did (RG_S_* above), this engine's groups. */
int rg_safety_summary(rg_engine *e, uint64_t *out /* RG_N_SUMMARY */);
/* Device addresses of the counter block (RG_N_COUNTERS x uint64) and of the
 * flag summary (2 x uint32, refreshed by rg_safety_flags) for callers that
 * run their own collectives. */
int rg_device_buffers(rg_engine *e, void **counters_dev, void **flag_summary_dev);

/* ---- the one cross-GPU exchange of the path: counters and safety summary, summed over the ranks ----------------
 * No counterpart in the reference (one process, one group per test).  Groups are sharded over GPUs by global id and
 * never exchange anything; what a multi-GPU job reports is the sum of every rank's counters (RG_C_*) and safety
 * summary (RG_S_*: groups per flag bit -- a bit is set somewhere in the job iff its count is non-zero).  The engine
 * packs the RG_N_COUNTERS + RG_N_SUMMARY words into one device buffer and reduces it in place with ONE ncclAllReduce
 * (uint64, sum) on its stream -- RCCL over xGMI, 248 bytes: latency-bound -- then copies the result out.  librccl.so
 * is loaded when first needed (RG_ERR_STATE if it cannot be); the library does not link against it.
 *   rg_comm_unique_id: ncclGetUniqueId on one rank; the caller carries the RG_COMM_ID_BYTES to the other ranks.
 *   rg_comm_init: ncclCommInitRank on the engine's device; collective over all n_ranks.  n_ranks == 1 needs no id.
 *   rg_allreduce_summary: nccl_comm = a caller-made ncclComm_t for the engine's device, or NULL for the communicator
 *     of rg_comm_init; with neither (a single-GPU job) the call just reports this engine's own words.  Either
 *     output pointer may be NULL.  Collective: every rank calls it once per reporting boundary, never per tick. */
#define RG_COMM_ID_BYTES 128
int rg_comm_unique_id(void *id_out);
int rg_comm_init(rg_engine *e, int32_t rank, int32_t n_ranks, const void *id);
int rg_comm_destroy(rg_engine *e);
int rg_allreduce_summary(rg_engine *e, void *nccl_comm, uint64_t *counters_out /* RG_N_COUNTERS */,
                         uint64_t *summary_out /* RG_N_SUMMARY */);

#if defined(RG_SECTION_COUNTS) || defined(RG_STAMPS)
/* Diagnostic builds of the library only (compiled with -DRG_SECTION_COUNTS or -DRG_STAMPS for the scripts under profiles/; the
 * shipped library does not export it): how often each section of the tick kernel ran and the shader cycles spent
 * there, 56 words.  No counterpart in the reference. */
int rg_debug_section_counts(rg_engine *e, uint64_t *out);
#endif

#ifdef __cplusplus
}
#endif
#endif /* RAFTGPU_H */
