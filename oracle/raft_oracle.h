/*
 * raft_oracle.h -- CPU restatement of the reference Raft peer, driven in
 * lockstep ticks.  TEST INFRASTRUCTURE ONLY: the checker the HIP engine is
 * compared against.  Nothing in the product path may include, link or call
 * this; only tests, the smoke check and the bench's CPU-baseline leg do.
 *
 * Parity pin: the reference ships no golden vectors for this path (its tests
 * are wall-clock driven and seeded from crypto/rand, src/raft/config.go:30-35,
 * 61-66) and its Go sources cannot be compiled in this image, so the oracle is
 * pinned by (a) the outcome properties the reference's tests assert
 * (src/raft/test_test.go:97-106,196-209,461,527,546; src/raft/config.go:
 * 177-193,317-320) and (b) hand-derived known answers from the reference
 * source, both in tests/.
 *
 * The state, config, field ids, flag bits and counters are those of
 * include/raftgpu.h so that dumps compare word for word.
 */
#ifndef RAFT_ORACLE_H
#define RAFT_ORACLE_H

#include "../include/raftgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_engine ro_engine;

ro_engine *ro_create(const rg_config *cfg);
void ro_destroy(ro_engine *e);
void ro_reset(ro_engine *e);
/* persist() copies the whole committed prefix on every call, as the
 * reference re-encodes it (raft.go:116-122), instead of only the new part.
 * Results are identical; only the cost differs. */
void ro_set_full_persist(ro_engine *e, int on);
/* TEST-ONLY departures from the reference's rules, to tell violations the reference's own rules cause from
 * violations a defect of the lockstep model would cause: with all three set, a schedule must end without a
 * safety flag.  Never set by parity tests (the HIP engine has no such switch).
 *   VOTE_LAST_LOG: RequestVote's up-to-date test and the candidate's LastLog* use the last log entry, not
 *                  commitIndex (raft.go:218-219, 582-583);
 *   PERSIST_LOG:   persist() stores the whole log, not logs[0:commitIndex+1] (raft.go:120), and readPersist
 *                  leaves commitIndex at 0 instead of len-1 (raft.go:148);
 *   RESTART:       a handler that makes the peer a follower clears the election loop's `restart`, so a round
 *                  of a later candidacy increments the term (raft.go:569-573, 660). */
#define RO_REPAIR_VOTE_LAST_LOG 1
#define RO_REPAIR_PERSIST_LOG   2
#define RO_REPAIR_RESTART       4
void ro_set_repaired_rules(ro_engine *e, int mask);

int ro_step(ro_engine *e, int32_t n_ticks);
int ro_step_mt(ro_engine *e, int32_t n_ticks, int32_t n_threads);
int64_t ro_now(ro_engine *e);

int ro_start(ro_engine *e, int32_t g, int32_t p, uint64_t cmd, int32_t *index, int32_t *term, int32_t *is_leader);
int ro_get_state(ro_engine *e, int32_t g, int32_t p, int32_t *term, int32_t *is_leader);
int ro_set_connected(ro_engine *e, int32_t g, int32_t p, int32_t on);
int ro_set_side(ro_engine *e, int32_t g, int32_t p, int32_t side);
int ro_crash(ro_engine *e, int32_t g, int32_t p);
int ro_kill(ro_engine *e, int32_t g, int32_t p);
int ro_restart(ro_engine *e, int32_t g, int32_t p);
int ro_set_unreliable(ro_engine *e, int32_t req_drop, int32_t reply_drop, int32_t reply_delay);
int ro_set_unreliable_group(ro_engine *e, int32_t g, int32_t req_drop, int32_t reply_drop, int32_t reply_delay);
int ro_reset_groups(ro_engine *e, const int32_t *groups, int64_t n);
int ro_set_schedule(ro_engine *e, const rg_sched_op *ops, int64_t n);

int ro_drain_applied(ro_engine *e, int32_t g, int32_t p, rg_apply_msg *buf, int64_t cap, int64_t *n_out);

int ro_rpc_request_vote(ro_engine *e, int32_t g, int32_t p, const rg_vote_args *a, rg_vote_reply *r);
int ro_rpc_append_entries(ro_engine *e, int32_t g, int32_t p, const rg_append_args *a,
                          const int32_t *entry_terms, const uint64_t *entry_cmds, rg_append_reply *r);

int ro_read_state(ro_engine *e, int32_t field, void *dst, int64_t dst_bytes);
int ro_write_state(ro_engine *e, int32_t field, const void *src, int64_t src_bytes);
int ro_state_hash(ro_engine *e, uint64_t *out, int64_t n_groups);
int ro_counters(ro_engine *e, uint64_t *out);
int ro_rpc_bytes(ro_engine *e, int64_t command_bytes, uint64_t *total);
int ro_safety_check(ro_engine *e);
int ro_safety_flags(ro_engine *e, uint32_t *out);
int ro_safety_summary(ro_engine *e, uint64_t *out /* RG_N_SUMMARY */);

int ro_snapshot_export(ro_engine *e, int32_t g, int32_t p, void *buf, int64_t cap, int64_t *size_out);
int ro_snapshot_import(ro_engine *e, int32_t g, int32_t p, const void *buf, int64_t size);

/* leader-side helpers exposed for known-answer tests */
int32_t ro_leader_backoff(ro_engine *e, int32_t g, int32_t leader, int32_t follower,
                          int32_t conflict_term, int32_t conflict_first_index);
int32_t ro_commit_rule(ro_engine *e, int32_t g, int32_t leader);

#ifdef __cplusplus
}
#endif
#endif
