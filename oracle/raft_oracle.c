/*
 * raft_oracle.c -- CPU restatement of the reference Raft peer (plain C).
 * TEST INFRASTRUCTURE ONLY (see raft_oracle.h): it is the checker, never the
 * product path.  Parity pin: the reference holds no golden vectors for this
 * path; pinned by the reference tests' outcome properties and hand-derived
 * known answers (tests/test_oracle_*.py).
 *
 * Each function follows the reference code it names, statement by statement.
 * What replaces goroutines, wall clock and the labrpc channel network is the
 * lockstep schedule below, which is one legal serialisation of the
 * reference's mutex-guarded steps:
 *
 *   tick t (10 ms), per group:
 *     1. control   : rolling-partition reshuffle, churn, client Start injection
 *     2. actors    : live peers in cyclic index order from a seeded per-tick
 *                    start (goroutine scheduling order is arbitrary in the
 *                    reference; a fixed order phase-locks its 50 ms loops):
 *                    the election loop iteration if its sleep is over, then
 *                    the leader loop (monitor) if its sleep is over.  An RPC
 *                    is a direct call of the target's handler, gated by the
 *                    network outcome (not delivered / handler ran, reply lost /
 *                    round trip), exactly in the reference's sequential
 *                    fan-out order with its break rules.  The handlers of a
 *                    round all run at the round's tick; the sender is busy for
 *                    the round's duration: per Call, the time until Call
 *                    returns (the unreliable net's 0..26 ms, labrpc.go:222,
 *                    for a reply and for a request or reply the net drops,
 *                    labrpc.go:226-229,273-275) or the 25 ms of the sender's
 *                    wait (raft.go:473,608), whichever is less; its loop
 *                    resumes after it.
 *     3. apply     : for p = 0..P-1: the apply loop iteration (10 ms period)
 *     4. check     : one-leader-per-term among connected peers
 *
 * Fixed-capacity arrays stand in for Go slices: truncation only lowers the
 * length and leaves the old words in place, as re-slicing a Go slice leaves
 * its backing array.  States in which the reference would index past
 * len(logs) raise RG_FLAG_REF_PANIC and read the stale words.
 */
#include "raft_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  /* Raft struct, src/raft/raft.go:57-86 */
  int32_t dead, term, vote, logLen, commit, applied;
  int32_t next[RG_MAX_PEERS], match[RG_MAX_PEERS];
  int32_t timer, state, hbTimer, stop, cnt;       /* timer, hbTimer in ms of simulated time */
  /* locals of the election goroutine, raft.go:543-547 */
  int32_t timeout, restart, elNext;
  /* the monitor goroutine, raft.go:387-394 */
  int32_t monAlive, first, monNext;
  /* the Persister blob, persister.go:14-18, decoded */
  int32_t snapHas, snapTerm, snapVote, snapLen, snapSync;
  /* engine-side: harness connectivity and seeded streams */
  int32_t connected, side;
  uint32_t rngCtr, roundSeq;
  uint64_t applyHash;
  int32_t *logTerm;  uint64_t *logCmd;
  int32_t *snapTerm_; uint64_t *snapCmd_;
  /* applyCh, raft.go:680-684: what the slot's peers sent, in order (a ring of cfg.apply_ring_cap messages) */
  rg_apply_msg *applyRing; uint32_t applyHead, applyDrained;
} peer_t;

typedef struct {
  peer_t peer[RG_MAX_PEERS];
  uint64_t key;          /* per-group stream key */
  int32_t startSeq, partNext, partEpoch, recMax;
  uint32_t flags;
  int32_t netOwn, reqDrop, replyDrop, replyDelay;   /* the group's own net rates (config.go:288-294), when netOwn */
  int32_t schedPos, schedRegs;                      /* scenario schedule: next op, four peer registers (4 bits each, 15 = none) */
  uint64_t *applyRec;    /* first command applied at each index, config.go:177-185 */
} group_t;

struct ro_engine {
  rg_config cfg;
  int P, L, G;
  int64_t now;
  int fullPersist;
  int repaired;          /* RO_REPAIR_* bits: test-only departures from the reference's rules (ro_set_repaired_rules) */
  rg_sched_op *sched; int schedN;
  group_t *groups;
  uint64_t counters[RG_N_COUNTERS];
};

/* per-thread counter block so that ro_step_mt needs no atomics */
typedef struct { uint64_t c[RG_N_COUNTERS]; } ctr_t;

/* ---------------- seeded streams ---------------------------------------- */
#define STREAM_ELECTION 1ull
#define STREAM_NET      2ull
#define STREAM_CMD      3ull
#define STREAM_PART     4ull
#define STREAM_CHURN    5ull
#define STREAM_SCHED    6ull

static inline uint64_t sm64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline uint64_t group_key(uint64_t seed, int64_t ggid) { return sm64(sm64(seed) ^ (uint64_t)ggid); }
static inline uint64_t stream_hash(uint64_t key, uint64_t stream, uint32_t a, uint32_t b) {
  return sm64(key ^ (stream << 56) ^ ((uint64_t)a << 32) ^ (uint64_t)b);
}

/* nextRandom(), raft.go:532-534 */
static int32_t next_random(group_t *g, int p) {
  peer_t *rf = &g->peer[p];
  uint64_t h = stream_hash(g->key, STREAM_ELECTION, (uint32_t)p, rf->rngCtr++);
  return (int32_t)((uint32_t)(h >> 32) % RG_TIMEOUT_SPAN_MS) + RG_TIMEOUT_BASE_MS;
}

/* time.Sleep never returns early and overshoots by scheduler slack: the
 * election loop's 50 ms sleep (raft.go:667) lasts 50..53 ms, seeded.  Without
 * it every loop of a group has exactly the same period and their phases lock. */
static int32_t sleep_slack(const group_t *g, int p, int32_t now) {
  uint32_t h = (uint32_t)g->key ^ ((uint32_t)p * 0x9E3779B1u) ^ ((uint32_t)now * 0x85EBCA6Bu);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return (int32_t)(h >> 30);
}

/* ---------------- persist / readPersist --------------------------------- */
/* persist(), raft.go:107-123: currentTerm, voteFor, logs[0:commitIndex+1]. */
static void persist(ro_engine *e, ctr_t *c, peer_t *rf) {
  int n = (e->repaired & RO_REPAIR_PERSIST_LOG) ? rf->logLen : rf->commit + 1;
  if (n > e->L) n = e->L;
  rf->snapTerm = rf->term;
  rf->snapVote = rf->vote;
  int from = e->fullPersist ? 0 : rf->snapSync;
  for (int i = from; i < n; i++) { rf->snapTerm_[i] = rf->logTerm[i]; rf->snapCmd_[i] = rf->logCmd[i]; }
  if (n > rf->snapSync) { c->c[RG_C_PERSIST_ENTRIES] += (uint64_t)(n - rf->snapSync); rf->snapSync = n; }
  rf->snapLen = n;
  rf->snapHas = 1;
  c->c[RG_C_PERSISTS]++;
}

/* Make(), raft.go:700-731, with readPersist(), raft.go:129-152, and the
 * start of election(), raft.go:538-547. */
static void make_peer(ro_engine *e, ctr_t *c, group_t *g, int p, int32_t now) {
  peer_t *rf = &g->peer[p];
  rf->dead = 0;
  rf->term = 0;
  rf->vote = -1;
  rf->logTerm[0] = 0; rf->logCmd[0] = 0;   /* Entry{Command:"init", Term:0} */
  rf->logLen = 1;
  rf->commit = 0;
  rf->applied = 0;
  rf->timer = now;
  rf->state = RG_FOLLOWER;
  rf->stop = 0;
  rf->cnt = 0;
  rf->hbTimer = now;
  if (rf->snapHas && rf->snapLen >= 1) {            /* readPersist */
    rf->term = rf->snapTerm;
    rf->vote = rf->snapVote;
    int n = rf->snapLen;
    int from = e->fullPersist ? 0 : rf->snapSync;
    for (int i = from; i < n; i++) { rf->logTerm[i] = rf->snapTerm_[i]; rf->logCmd[i] = rf->snapCmd_[i]; }
    if (n > rf->snapSync) rf->snapSync = n;
    rf->logLen = n;
    persist(e, c, rf);                              /* :147, commitIndex is still 0 */
    if (!(e->repaired & RO_REPAIR_PERSIST_LOG)) rf->commit = rf->logLen - 1;   /* :148 */
  }
  rf->monAlive = 0; rf->first = 0; rf->monNext = 0;
  rf->restart = 0;
  rf->timeout = 0;
  if (!(rf->term == 0 && p == 0)) rf->timeout = next_random(g, p);   /* :545-547 */
  rf->elNext = now;                                 /* first loop iteration right away */
}

/* GetState(), raft.go:90-100 */
static int is_leader(const peer_t *rf) { return !rf->dead && !rf->stop && rf->state == RG_LEADER; }

/* ---------------- RPC handlers ------------------------------------------ */
/* RequestVote(), raft.go:200-232 */
static void request_vote(ro_engine *e, ctr_t *c, peer_t *rf, int32_t now, const rg_vote_args *a, rg_vote_reply *r) {
  if (rf->dead) return;
  if (a->term < rf->term) {
    r->granted = 0;
    r->term = rf->term;
  } else {
    if (a->term > rf->term) { rf->term = a->term; rf->vote = -1; persist(e, c, rf); }
    r->term = rf->term;
    rf->state = RG_FOLLOWER;                                      /* :216 */
    if (e->repaired & RO_REPAIR_RESTART) rf->restart = 0;
    int32_t myLastIndex = rf->commit;
    if (e->repaired & RO_REPAIR_VOTE_LAST_LOG) myLastIndex = rf->logLen - 1;
    int32_t myLastTerm = rf->logTerm[myLastIndex];                /* logs[commitIndex].Term */
    if ((rf->vote == -1 || rf->vote == a->candidate) &&
        (a->last_log_term > myLastTerm || (a->last_log_term == myLastTerm && a->last_log_index >= myLastIndex))) {
      r->granted = 1;
      rf->vote = a->candidate;
      persist(e, c, rf);
      rf->timer = now;
    } else {
      r->granted = 0;
    }
  }
}

/* AppendEntries(), raft.go:270-321.  Entries are (terms, cmds)[0..n). */
static void append_entries(ro_engine *e, ctr_t *c, group_t *g, peer_t *rf, int32_t now, const rg_append_args *a,
                           const int32_t *et, const uint64_t *ec, rg_append_reply *r) {
  if (rf->dead) return;
  if (a->term < rf->term) {
    r->success = 0;
    r->term = rf->term;
  } else {
    rf->state = RG_FOLLOWER;
    if (e->repaired & RO_REPAIR_RESTART) rf->restart = 0;
    rf->vote = a->leader;                                         /* :281 */
    if (rf->term < a->term) { rf->term = a->term; persist(e, c, rf); }
    r->term = rf->term;
    rf->timer = now;
    if (a->prev_log_index + 1 <= rf->logLen && rf->logTerm[a->prev_log_index] == a->prev_log_term) {
      r->success = 1;
      int base = a->prev_log_index + 1;                           /* logs[0:Prev+1] ++ Entries */
      for (int i = 0; i < a->n_entries; i++) { rf->logTerm[base + i] = et[i]; rf->logCmd[base + i] = ec[i]; }
      if (a->n_entries > 0 && base < rf->snapSync) rf->snapSync = base;
      rf->logLen = base + a->n_entries;
      if (a->leader_commit > rf->commit) {
        if (rf->logLen - 1 < a->leader_commit) rf->commit = rf->logLen - 1;
        else rf->commit = a->leader_commit;
      }
      if (rf->commit > rf->logLen - 1) g->flags |= RG_FLAG_REF_PANIC;
      persist(e, c, rf);
      c->c[RG_C_ENTRIES_REPLICATED] += (uint64_t)a->n_entries;
    } else {
      r->success = 0;
      if (a->prev_log_index + 1 > rf->logLen) {
        r->conflict_term = -1;
        r->conflict_first_index = rf->logLen;
      } else {
        r->conflict_term = rf->logTerm[a->prev_log_index];
        int index = a->prev_log_index;
        while (index >= 0 && rf->logTerm[index] == r->conflict_term) index--;
        r->conflict_first_index = index + 1;
      }
      c->c[RG_C_AE_CONFLICTS]++;
    }
  }
}

/* ---------------- network (labrpc.go:216-307) --------------------------- */
enum { NET_LOST = 0, NET_REPLY_LOST = 1, NET_OK = 2 };

/* *cost: how long the sender's select (raft.go:469-474, 605-609) waits for this Call: until Call returns, true
 * or false, or 25 ms, whichever comes first.  A request or reply the net drops makes Call return false right
 * after the net's short delay (labrpc.go:226-229, 273-275); a Call to an unreachable peer returns false after
 * 0..6999 ms (labrpc.go:290-305; the tester sets longDelays, config.go:82), i.e. before the 25 ms are over in 25
 * of 7000 Calls; a long-reordered reply comes after 200 ms or more.
 * A peer is unreachable when an end is disabled, the sides differ, or its server was deleted (crash1,
 * config.go:99-127: dead == 1).  A peer that was only Kill()ed (raft.go:377-380: dead == 2) still has its server
 * registered: the Call is delivered, the handler returns at once (raft.go:203,273) and Call returns true with
 * the reply untouched. */
static inline int32_t permille_of(uint32_t field16) { return (int32_t)((field16 * 1000u) >> 16); }

static int net_outcome(const ro_engine *e, const group_t *g, int s, int j, uint32_t seq, int32_t *cost, int *wire) {
  const peer_t *a = &g->peer[s], *b = &g->peer[j];
  *cost = RG_RPC_WAIT_MS;
  *wire = 0;                      /* does the reply travel (and count its bytes, labrpc.go:286-290), in time or not */
  if (!(a->connected && b->connected && a->side == b->side) || b->dead == 1) {
    uint64_t h = stream_hash(g->key, STREAM_NET, ((uint32_t)s << 8) | (uint32_t)j, seq);
    int32_t ms = (int32_t)(((h >> 32) * 7000ull) >> 32);            /* labrpc.go:296, 0..6999 */
    if (ms < RG_RPC_WAIT_MS) *cost = ms;
    return NET_LOST;
  }
  const int32_t reqDrop = g->netOwn ? g->reqDrop : e->cfg.req_drop_permille;
  const int32_t replyDrop = g->netOwn ? g->replyDrop : e->cfg.reply_drop_permille;
  const int32_t replyDelay = g->netOwn ? g->replyDelay : e->cfg.reply_delay_permille;
  if (reqDrop == 0 && replyDrop == 0 && replyDelay == 0) { *cost = 0; *wire = 1; return NET_OK; }
  /* one hash per Call: 16 bits each for the two drop draws and the reordering draw, 8 bits for the delay */
  uint64_t h = stream_hash(g->key, STREAM_NET, ((uint32_t)s << 8) | (uint32_t)j, seq);
  int32_t d = (int32_t)((((h >> 48) & 0xFF) * RG_NET_DELAY_SPAN_MS) >> 8);                   /* :222, 0..26 */
  int32_t back = d < RG_RPC_WAIT_MS ? d : RG_RPC_WAIT_MS;           /* Call returns after d ms; the select gives up at 25 */
  if (permille_of((uint32_t)(h & 0xFFFF)) < reqDrop) { *cost = back; return NET_LOST; }   /* :226-229 */
  *wire = !(permille_of((uint32_t)((h >> 16) & 0xFFFF)) < replyDrop);           /* :273 */
  if (!*wire) { *cost = back; return NET_REPLY_LOST; }               /* :273-275 */
  if (d >= RG_RPC_WAIT_MS) return NET_REPLY_LOST;                    /* handler runs, the sender has given up */
  if (permille_of((uint32_t)((h >> 32) & 0xFFFF)) < replyDelay) return NET_REPLY_LOST;  /* :276-285 */
  *cost = d;
  return NET_OK;
}

/* ---------------- Start (raft.go:344-364) ------------------------------- */
static void start_cmd(ro_engine *e, ctr_t *c, group_t *g, int p, uint64_t cmd, int32_t *index, int32_t *term, int32_t *leader) {
  peer_t *rf = &g->peer[p];
  *index = -1;
  *term = rf->term;
  *leader = is_leader(rf);
  if (*leader) {
    if (rf->logLen >= e->L) { g->flags |= RG_FLAG_LOG_OVERFLOW; *leader = 0; return; }
    *index = rf->logLen;
    rf->logTerm[rf->logLen] = rf->term;
    rf->logCmd[rf->logLen] = cmd;
    if (rf->logLen < rf->snapSync) rf->snapSync = rf->logLen;
    rf->logLen++;
    persist(e, c, rf);
    c->c[RG_C_ENTRIES_STARTED]++;
  }
}

/* ---------------- election loop iteration (raft.go:548-668) ------------- */
static void election_step(ro_engine *e, ctr_t *c, group_t *g, int me, int32_t now) {
  peer_t *rf = &g->peer[me];
  const int l = e->P;
  int32_t state = rf->state;
  int32_t timer = rf->timer;
  switch (state) {
  case RG_FOLLOWER:
    if (now - timer > rf->timeout) {                               /* :557 */
      rf->state = RG_CANDIDATE;
      rf->timeout = next_random(g, me);
      c->c[RG_C_TIMEOUTS]++;
    }
    break;
  case RG_CANDIDATE: {
    rf->timeout = RG_CANDIDATE_WINDOW_MS;
    rf->timer = now;
    if (!rf->restart) rf->term = rf->term + 1; else rf->restart = 0;
    rf->vote = me;
    persist(e, c, rf);
    rg_vote_args args = { rf->term, me, rf->commit, rf->logTerm[rf->commit] };   /* :579-584 */
    if (e->repaired & RO_REPAIR_VOTE_LAST_LOG) { args.last_log_index = rf->logLen - 1; args.last_log_term = rf->logTerm[rf->logLen - 1]; }
    rg_vote_reply reply = { rf->term, 0 };
    persist(e, c, rf);
    c->c[RG_C_RV_ROUNDS]++;
    uint32_t seq = rf->roundSeq++;
    int voteCnt = 1;
    int32_t busy = 0, cost;
    for (int i = 0; i < l && state == RG_CANDIDATE; i++) {
      state = rf->state;
      if (i != me) {
        int onWire;
        int out = net_outcome(e, g, me, i, seq, &cost, &onWire);
        busy += cost;
        c->c[RG_C_RPC_SENT]++;
        if (e->cfg.account_bytes) c->c[RG_C_RPC_FIXED_BYTES] += RG_VOTE_ARGS_BYTES + (onWire ? RG_VOTE_REPLY_BYTES : 0);
        int ok = 0;
        if (out != NET_LOST) {
          rg_vote_reply wire = { 0, 0 };
          request_vote(e, c, &g->peer[i], now, &args, &wire);
          c->c[RG_C_RPC_DELIVERED]++;
          if (out == NET_OK) {           /* gob leaves zero fields of the shared reply alone, labgob.go:116-121 */
            ok = 1;
            if (wire.term != 0) reply.term = wire.term;
            if (wire.granted) reply.granted = 1;
            c->c[RG_C_RPC_REPLIED]++;
          }
        }
        if (ok) {
          if (reply.granted) { voteCnt++; reply.granted = 0; }
          if (rf->term != args.term) break;
          if (reply.term > rf->term) {                             /* :622-627 */
            rf->state = RG_FOLLOWER;
            rf->vote = -1;
            persist(e, c, rf);
            c->c[RG_C_STEPDOWNS]++;
            break;
          }
        }
      }
    }
    state = rf->state;
    if (!rf->dead) {
      if (state == RG_FOLLOWER) {
        rf->timeout = next_random(g, me);                          /* :639 */
      } else if (state == RG_CANDIDATE) {
        if (now + busy - rf->timer < rf->timeout && 2 * voteCnt > l) {   /* :641 */
          rf->stop = 0;
          rf->cnt = 0;
          rf->state = RG_LEADER;
          rf->hbTimer = now + busy;
          for (int i = 0; i < l; i++) rf->next[i] = rf->logLen;
          for (int i = 0; i < l; i++) rf->match[i] = 0;
          rf->monAlive = 1; rf->first = 1; rf->monNext = now + busy; /* go monitor(rf) */
          c->c[RG_C_ELECTIONS_WON]++;
        } else {
          rf->restart = 1;
        }
      }
    }
    state = RG_CANDIDATE;   /* the switch ran on the value loaded at the top */
    rf->elNext = now + busy + RG_ELECTION_PERIOD_MS + sleep_slack(g, me, now);
    return;
  }
  default: break;
  }
  rf->elNext = now + RG_ELECTION_PERIOD_MS + (state == RG_LEADER ? RG_ELECTION_LEADER_EXTRA_MS : 0) + sleep_slack(g, me, now);
}

/* commit rule, raft.go:400-416.  Returns 1 when commitIndex advanced. */
static int commit_rule(ro_engine *e, ctr_t *c, peer_t *rf, int me) {
  const int l = e->P;
  int N = rf->logLen - 1;
  for (; rf->state == RG_LEADER && N > rf->commit && rf->logTerm[N] == rf->term; N--) {
    int cnt = 1;
    for (int i = 0; i < l; i++) if (i != me && rf->match[i] >= N) cnt++;
    if (2 * cnt > l && rf->logTerm[N] == rf->term) {
      c->c[RG_C_ENTRIES_COMMITTED] += (uint64_t)(N - rf->commit);
      rf->commit = N;
      persist(e, c, rf);
      return 1;
    }
  }
  return 0;
}

/* leader's handling of a failed AppendEntries reply, raft.go:493-505 */
static int32_t leader_backoff(const peer_t *rf, int i, int32_t conflictTerm, int32_t conflictFirst) {
  if (conflictTerm >= 0) {
    int index = rf->next[i] - 1;
    while (rf->logTerm[index] > conflictTerm) index--;
    if (rf->logTerm[index] == conflictTerm) return index + 1;
    return conflictFirst;
  }
  return conflictFirst;
}

/* leader loop, raft.go:395-529, run until it would sleep or exit */
static void monitor_step(ro_engine *e, ctr_t *c, group_t *g, int me, int32_t now) {
  peer_t *rf = &g->peer[me];
  const int l = e->P;
  for (;;) {
    if (rf->dead) { rf->monAlive = 0; rf->monNext = 0; return; }
    int isleader = is_leader(rf);
    int active = 1;
    if (!isleader) { rf->monAlive = 0; rf->monNext = 0; return; }   /* :526-528 */
    if (commit_rule(e, c, rf, me)) rf->first = 1;
    int32_t timer = rf->hbTimer;
    int32_t logLen = rf->logLen;
    int32_t commitIndex = rf->commit;
    int32_t since = now - timer;
    if (rf->first || since >= RG_HB_IDLE_MS || (commitIndex < logLen - 1 && since >= RG_HB_BACKLOG_MS)) {   /* :421 */
      rf->first = 0;
      c->c[RG_C_AE_ROUNDS]++;
      uint32_t seq = rf->roundSeq++;
      int32_t busy = 0, cost;
      for (int i = 0; i < l; i++) {
        if (!is_leader(rf)) break;                                  /* :424-427 */
        if (i == me) continue;
        rg_append_args args;
        if (rf->next[i] <= logLen) {
          args.term = rf->term;
          args.leader = me;
          args.prev_log_index = rf->next[i] - 1;
          args.prev_log_term = rf->logTerm[rf->next[i] - 1];
          args.n_entries = logLen - rf->next[i];
          args.entry_off = rf->next[i];
          args.leader_commit = rf->commit;
        } else {
          break;                                                    /* :451-454 */
        }
        rg_append_reply reply = { 0, 0, 0, 0 };
        int onWire;
        int out = net_outcome(e, g, me, i, seq, &cost, &onWire);
        busy += cost;
        c->c[RG_C_RPC_SENT]++;
        if (e->cfg.account_bytes) {
          c->c[RG_C_RPC_FIXED_BYTES] += RG_APPEND_ARGS_BYTES + (onWire ? RG_APPEND_REPLY_BYTES : 0);
          c->c[RG_C_AE_ENTRIES_SENT] += (uint64_t)args.n_entries;
        }
        int ok = 0;
        if (out != NET_LOST) {
          append_entries(e, c, g, &g->peer[i], now, &args, rf->logTerm + args.entry_off, rf->logCmd + args.entry_off, &reply);
          c->c[RG_C_RPC_DELIVERED]++;
          c->c[RG_C_AE_DELIVERED]++;
          if (out == NET_OK) { ok = 1; c->c[RG_C_RPC_REPLIED]++; }
        }
        if (ok) {
          active++;
          if (rf->term != args.term) break;
          else if (reply.term > rf->term) {                         /* :482-487 */
            rf->state = RG_FOLLOWER;
            rf->term = reply.term;
            persist(e, c, rf);
            c->c[RG_C_STEPDOWNS]++;
            break;
          } else {
            if (reply.success) {
              rf->next[i] = logLen;
              rf->match[i] = rf->next[i] - 1;
            } else {
              rf->next[i] = leader_backoff(rf, i, reply.conflict_term, reply.conflict_first_index);
            }
          }
        }
      }
      if (active == 1 && rf->cnt <= 3) {                            /* :514-520 */
        rf->cnt++;
        if (rf->cnt > 3) rf->stop = 1;
      }
      rf->hbTimer = now + busy;                                     /* :521 */
      if (busy > 0) { rf->monNext = now + busy; return; }           /* the loop goes on once the waits are over */
    } else {
      rf->monNext = now + RG_MONITOR_POLL_MS;                       /* :524 */
      return;
    }
  }
}

/* apply loop iteration, raft.go:676-686, and the harness reader, config.go:170-194 */
static void apply_step(ro_engine *e, ctr_t *c, group_t *g, int p) {
  peer_t *rf = &g->peer[p];
  while (rf->commit > rf->applied) {
    rf->applied++;
    int idx = rf->applied;
    if (idx >= rf->logLen) g->flags |= RG_FLAG_REF_PANIC;
    if (idx >= e->L) idx = e->L - 1;
    uint64_t cmd = rf->logCmd[idx];
    if (rf->applyRing) {           /* rf.applyCh <- ApplyMsg{true, logs[lastApplied].Command, lastApplied} */
      rg_apply_msg *m = &rf->applyRing[rf->applyHead++ % (uint32_t)e->cfg.apply_ring_cap];
      m->command_index = rf->applied; m->command = cmd;
    }
    {  /* apply digest: two independent 32-bit sums of a hash of (index, command), so that entries can be folded in any order */
      uint64_t wd = sm64(cmd ^ ((uint64_t)(uint32_t)rf->applied << 40));
      uint32_t lo = (uint32_t)rf->applyHash + (uint32_t)wd, hi = (uint32_t)(rf->applyHash >> 32) + (uint32_t)(wd >> 32);
      rf->applyHash = ((uint64_t)hi << 32) | lo;
    }
    if (rf->applied <= g->recMax) {
      if (g->applyRec[idx] != cmd) g->flags |= RG_FLAG_STATE_MACHINE;
    } else {
      if (rf->applied != g->recMax + 1) g->flags |= RG_FLAG_APPLY_ORDER;
      g->applyRec[idx] = cmd;
      g->recMax = rf->applied;
    }
    c->c[RG_C_ENTRIES_APPLIED]++;
  }
}

/* ---------------- harness verbs (config.go:99-278) ----------------------- */
static void crash1(group_t *g, int p) {
  g->peer[p].connected = 0;
  g->peer[p].dead = 1;
}
/* rf.Kill(), raft.go:377-380: the dead flag alone.  The peer's loops exit (:395,:548,:674), its handlers return at
 * once (:203,:273); its server stays registered with the net, so Calls to it still come back true (dead == 2).
 * A peer whose server was already deleted stays that way. */
static void kill1(group_t *g, int p) {
  if (!g->peer[p].dead) g->peer[p].dead = 2;
}
static void start1(ro_engine *e, ctr_t *c, group_t *g, int p, int32_t now) {
  crash1(g, p);
  make_peer(e, c, g, p, now);
}

/* ---------------- scenario schedule: the tester's verbs, one op ----------- */
static int sched_resolve(const ro_engine *e, const group_t *g, int sel) {   /* -1 nobody, -2 everybody */
  if (sel & 0x100) {
    int base = (g->schedRegs >> (4 * ((sel >> 4) & 3))) & 15;
    return base == 15 ? -1 : (base + (sel & 15)) % e->P;
  }
  return sel == RG_SEL_ALL ? -2 : sel;
}
static void sched_set_reg(group_t *g, int r, int val) {
  int sh = 4 * (r & 3);
  g->schedRegs = (g->schedRegs & ~(15 << sh)) | (val << sh);
}
static void sched_op(ro_engine *e, ctr_t *c, group_t *g, const rg_sched_op *op, int pos, int32_t tick) {
  const int P = e->P;
  const int tgt = sched_resolve(e, g, op->a);
  int32_t idx, term, ok;
  switch (op->op) {
  case RG_OP_CONNECT:    for (int p = 0; p < P; p++) if (tgt == p || tgt == -2) g->peer[p].connected = 1; break;   /* config.go:235-256 */
  case RG_OP_DISCONNECT: for (int p = 0; p < P; p++) if (tgt == p || tgt == -2) g->peer[p].connected = 0; break;   /* config.go:258-278 */
  case RG_OP_CRASH:      for (int p = 0; p < P; p++) if (tgt == p || tgt == -2) crash1(g, p); break;
  case RG_OP_KILL:       for (int p = 0; p < P; p++) if (tgt == p || tgt == -2) kill1(g, p); break;                     /* raft.go:377-380 */
  case RG_OP_RESTART:    for (int p = 0; p < P; p++) if (tgt == p || tgt == -2) start1(e, c, g, p, tick * RG_TICK_MS); break;
  case RG_OP_RESTART_IF_FEWER: {                                            /* test_test.go:718-725 */
    int alive = 0;
    for (int p = 0; p < P; p++) alive += !g->peer[p].dead;
    for (int p = 0; p < P; p++)
      if ((tgt == p || tgt == -2) && g->peer[p].dead && alive < op->b) { start1(e, c, g, p, tick * RG_TICK_MS); g->peer[p].connected = 1; }
    break;
  }
  case RG_OP_CONNECT_IF_FEWER: {                                            /* test_test.go:805-811 */
    int up = 0;
    for (int p = 0; p < P; p++) up += g->peer[p].connected != 0;
    for (int p = 0; p < P; p++) if ((tgt == p || tgt == -2) && up < op->b) g->peer[p].connected = 1;
    break;
  }
  case RG_OP_START:                                                         /* rf.Start on one peer */
    if (tgt >= 0 && !g->peer[tgt].dead) start_cmd(e, c, g, tgt, stream_hash(g->key, STREAM_CMD, 1, (uint32_t)op->b), &idx, &term, &ok);
    break;
  case RG_OP_START_ANY:                                                     /* config.go:436-451 */
    for (int p = 0; p < P; p++) {
      peer_t *rf = &g->peer[p];
      if (rf->dead || !rf->connected || !is_leader(rf)) continue;
      start_cmd(e, c, g, p, stream_hash(g->key, STREAM_CMD, 1, (uint32_t)op->b), &idx, &term, &ok);
      if (ok) break;
    }
    break;
  case RG_OP_START_ALL: {                                                   /* test_test.go:697-703 */
    int last = 15;
    for (int p = 0; p < P; p++) {
      if (g->peer[p].dead) continue;
      start_cmd(e, c, g, p, stream_hash(g->key, STREAM_CMD, 1, (uint32_t)op->b), &idx, &term, &ok);
      if (ok) last = p;
    }
    sched_set_reg(g, op->a, last);
    break;
  }
  case RG_OP_PICK_LEADER: {                                                 /* config.go:302-332 */
    int best = 15; int32_t bt = -1;
    for (int p = 0; p < P; p++) {
      const peer_t *rf = &g->peer[p];
      if (rf->connected && is_leader(rf) && rf->term > bt) { bt = rf->term; best = p; }
    }
    sched_set_reg(g, op->a, best);
    break;
  }
  case RG_OP_PICK_RANDOM:
    sched_set_reg(g, op->a, (int)((uint32_t)(stream_hash(g->key, STREAM_SCHED, (uint32_t)pos, (uint32_t)op->b) >> 32) % (uint32_t)P));
    break;
  case RG_OP_PICK_OTHER: {
    int base = sched_resolve(e, g, op->b), val = 15;
    int from = base < 0 ? 0 : base + 1;
    for (int q = 0; q < P && val == 15; q++) {
      int p = (from + q) % P;
      if (p != base && g->peer[p].connected && !g->peer[p].dead) val = p;
    }
    sched_set_reg(g, op->a, val);
    break;
  }
  default: break;
  }
}

/* ---------------- control phase ----------------------------------------- */
static void control_step(ro_engine *e, ctr_t *c, group_t *g, int32_t now /* tick */) {
  const rg_config *cf = &e->cfg;
  const int P = e->P;
  /* the scenario schedule, before anything else of the tick */
  while (g->schedPos < e->schedN && e->sched[g->schedPos].tick <= now) {
    sched_op(e, c, g, &e->sched[g->schedPos], g->schedPos, now);
    g->schedPos++;
  }
  /* rolling partitions, kvraft/test_test.go:124-143 + kvraft/config.go:174-186 */
  if (cf->part_period > 0 && now == g->partNext) {
    uint64_t h = stream_hash(g->key, STREAM_PART, (uint32_t)g->partEpoch, 0);
    for (int p = 0; p < P; p++) g->peer[p].side = (int32_t)((h >> p) & 1);
    g->partNext = now + cf->part_period + (int32_t)(((h >> 32) * (uint64_t)(uint32_t)cf->part_jitter) >> 32);
    g->partEpoch++;
  }
  /* churn, test_test.go:893-919 */
  if (cf->churn_period > 0 && now > 0 && now % cf->churn_period == 0) {
    uint32_t it = (uint32_t)(now / cf->churn_period);
    uint64_t h1 = stream_hash(g->key, STREAM_CHURN, it, 0);
    uint64_t h2 = stream_hash(g->key, STREAM_CHURN, it, 1);
    uint64_t h3 = stream_hash(g->key, STREAM_CHURN, it, 2);
    if ((int32_t)((h1 & 0xFFFFF) % 1000) < cf->churn_disconnect_permille) {
      int i = (int)((h1 >> 32) % (uint64_t)P);
      g->peer[i].connected = 0;
    }
    if ((int32_t)((h2 & 0xFFFFF) % 1000) < cf->churn_restart_permille) {
      int i = (int)((h2 >> 32) % (uint64_t)P);
      if (g->peer[i].dead) start1(e, c, g, i, now * RG_TICK_MS);
      g->peer[i].connected = 1;
    }
    if ((int32_t)((h3 & 0xFFFFF) % 1000) < cf->churn_crash_permille) {
      int i = (int)((h3 >> 32) % (uint64_t)P);
      if (!g->peer[i].dead) crash1(g, i);
    }
  }
  /* client Starts, config.go:436-451 / test_test.go:697-703 */
  if (cf->start_period > 0 && now >= cf->start_first_tick &&
      (now - cf->start_first_tick) % cf->start_period == 0) {
    for (int p = 0; p < P; p++) {
      if (g->startSeq >= cf->start_max) break;
      peer_t *rf = &g->peer[p];
      if (rf->dead) continue;
      if (cf->start_mode == RG_START_ONE && !rf->connected) continue;
      if (!is_leader(rf)) continue;
      uint64_t cmd = stream_hash(g->key, STREAM_CMD, 0, (uint32_t)g->startSeq);
      int32_t idx, term, ok;
      start_cmd(e, c, g, p, cmd, &idx, &term, &ok);
      if (ok) {
        g->startSeq++;
        if (cf->start_mode == RG_START_ONE) break;
      }
    }
  }
}

/* checkOneLeader's invariant, config.go:307-320 */
static void check_leaders(ro_engine *e, group_t *g) {
  for (int a = 0; a < e->P; a++) {
    const peer_t *x = &g->peer[a];
    if (!x->connected || !is_leader(x)) continue;
    for (int b = a + 1; b < e->P; b++) {
      const peer_t *y = &g->peer[b];
      if (y->connected && is_leader(y) && y->term == x->term) g->flags |= RG_FLAG_ELECTION_SAFETY;
    }
  }
}

/* first peer of this tick's actor order */
static int turn_start(const group_t *g, int32_t tick, int P) {
  uint32_t h = (uint32_t)tick * 0x9E3779B1u + (uint32_t)g->key;
  h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
  return (int)(((h >> 16) * (uint32_t)P) >> 16);
}

static void tick_group(ro_engine *e, ctr_t *c, group_t *g, int32_t tick) {
  const int32_t now = tick * RG_TICK_MS;
  control_step(e, c, g, tick);
  const int start = turn_start(g, tick, e->P);
  for (int q = 0; q < e->P; q++) {
    int p = start + q; if (p >= e->P) p -= e->P;
    peer_t *rf = &g->peer[p];
    if (rf->dead) continue;
    if (rf->elNext <= now) election_step(e, c, g, p, now);
    if (!rf->dead && rf->monAlive && rf->monNext <= now) monitor_step(e, c, g, p, now);
  }
  for (int p = 0; p < e->P; p++) if (!g->peer[p].dead) apply_step(e, c, g, p);
  check_leaders(e, g);
}

/* ---------------- engine lifecycle -------------------------------------- */
/* make_config, config.go:60-96: fresh peers with empty Persisters, booted at tick nowTick */
static void init_group(ro_engine *e, group_t *g, int64_t ggid, int32_t nowTick) {
  ctr_t scratch; memset(&scratch, 0, sizeof scratch);
  g->key = group_key(e->cfg.seed, ggid);
  g->startSeq = 0; g->partEpoch = 0; g->recMax = 0; g->flags = 0;
  g->netOwn = 0; g->reqDrop = g->replyDrop = g->replyDelay = 0;
  g->schedRegs = 0xFFFF;
  if (nowTick == 0) g->schedPos = 0;
  g->partNext = e->cfg.part_period > 0 ? nowTick + e->cfg.part_period : 0;
  memset(g->applyRec, 0, sizeof(uint64_t) * (size_t)e->L);
  for (int p = 0; p < e->P; p++) {
    peer_t *rf = &g->peer[p];
    int32_t *lt = rf->logTerm, *st = rf->snapTerm_; uint64_t *lc = rf->logCmd, *sc = rf->snapCmd_;
    rg_apply_msg *ring = rf->applyRing;
    const uint32_t head = nowTick ? rf->applyHead : 0;    /* a reset group gets new apply channels: nothing pending */
    memset(rf, 0, sizeof *rf);
    rf->logTerm = lt; rf->snapTerm_ = st; rf->logCmd = lc; rf->snapCmd_ = sc; rf->applyRing = ring;
    rf->applyHead = rf->applyDrained = head;
    memset(lt, 0, sizeof(int32_t) * (size_t)e->L); memset(st, 0, sizeof(int32_t) * (size_t)e->L);
    memset(lc, 0, sizeof(uint64_t) * (size_t)e->L); memset(sc, 0, sizeof(uint64_t) * (size_t)e->L);
    rf->connected = 1; rf->side = 0;
    rf->snapSync = 1;          /* entry 0 is the same sentinel in every array */
    make_peer(e, &scratch, g, p, nowTick * RG_TICK_MS);
  }
}

static int valid_config(const rg_config *c) {
  if (!c) return 0;
  if (c->n_groups < 1 || c->n_peers < 2 || c->n_peers > RG_MAX_PEERS || c->log_cap < 2) return 0;
  if (c->start_period < 0 || c->part_period < 0 || c->churn_period < 0) return 0;
  if (c->start_period > 0 && (c->start_max < 0 || c->start_max + 1 > c->log_cap)) return 0;
  if (c->apply_ring_cap < 0) return 0;
  return 1;
}

ro_engine *ro_create(const rg_config *cfg) {
  if (!valid_config(cfg)) return NULL;
  ro_engine *e = (ro_engine *)calloc(1, sizeof *e);
  if (!e) return NULL;
  e->cfg = *cfg; e->P = cfg->n_peers; e->L = cfg->log_cap; e->G = cfg->n_groups;
  e->groups = (group_t *)calloc((size_t)e->G, sizeof(group_t));
  if (!e->groups) { free(e); return NULL; }
  for (int g = 0; g < e->G; g++) {
    group_t *gr = &e->groups[g];
    gr->applyRec = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)e->L);
    for (int p = 0; p < e->P; p++) {
      peer_t *rf = &gr->peer[p];
      rf->logTerm = (int32_t *)malloc(sizeof(int32_t) * (size_t)e->L);
      rf->snapTerm_ = (int32_t *)malloc(sizeof(int32_t) * (size_t)e->L);
      rf->logCmd = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)e->L);
      rf->snapCmd_ = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)e->L);
      rf->applyRing = cfg->apply_ring_cap > 0 ? (rg_apply_msg *)calloc((size_t)cfg->apply_ring_cap, sizeof(rg_apply_msg)) : NULL;
    }
  }
  ro_reset(e);
  return e;
}

void ro_reset(ro_engine *e) {
  e->now = 0;
  memset(e->counters, 0, sizeof e->counters);
  for (int g = 0; g < e->G; g++) init_group(e, &e->groups[g], e->cfg.group_base + g, 0);
}

void ro_destroy(ro_engine *e) {
  if (!e) return;
  for (int g = 0; g < e->G; g++) {
    group_t *gr = &e->groups[g];
    free(gr->applyRec);
    for (int p = 0; p < e->P; p++) {
      free(gr->peer[p].logTerm); free(gr->peer[p].snapTerm_); free(gr->peer[p].logCmd); free(gr->peer[p].snapCmd_);
      free(gr->peer[p].applyRing);
    }
  }
  free(e->groups);
  free(e->sched);
  free(e);
}

void ro_set_full_persist(ro_engine *e, int on) { e->fullPersist = on; }
void ro_set_repaired_rules(ro_engine *e, int mask) { e->repaired = mask; }
int64_t ro_now(ro_engine *e) { return e->now; }

static void add_counters(ro_engine *e, const ctr_t *c) { for (int i = 0; i < RG_N_COUNTERS; i++) e->counters[i] += c->c[i]; }

int ro_step(ro_engine *e, int32_t n_ticks) {
  if (!e || n_ticks < 0) return RG_ERR_ARG;
  ctr_t c; memset(&c, 0, sizeof c);
  /* groups are independent: advance each one through all ticks in turn */
  for (int g = 0; g < e->G; g++)
    for (int32_t t = 0; t < n_ticks; t++) tick_group(e, &c, &e->groups[g], (int32_t)(e->now + t));
  c.c[RG_C_GROUP_TICKS] += (uint64_t)e->G * (uint64_t)n_ticks;
  add_counters(e, &c);
  e->now += n_ticks;
  return RG_OK;
}

typedef struct { ro_engine *e; int g0, g1; int32_t n; ctr_t c; } job_t;
static void *mt_worker(void *arg) {
  job_t *j = (job_t *)arg;
  for (int g = j->g0; g < j->g1; g++)
    for (int32_t t = 0; t < j->n; t++) tick_group(j->e, &j->c, &j->e->groups[g], (int32_t)(j->e->now + t));
  return NULL;
}

int ro_step_mt(ro_engine *e, int32_t n_ticks, int32_t n_threads) {
  if (!e || n_ticks < 0 || n_threads < 1) return RG_ERR_ARG;
  if (n_threads > e->G) n_threads = e->G;
  if (n_threads > 256) n_threads = 256;
  job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  if (!jobs || !th) { free(jobs); free(th); return RG_ERR_NOMEM; }
  for (int i = 0; i < n_threads; i++) {
    jobs[i].e = e; jobs[i].n = n_ticks;
    jobs[i].g0 = (int)((int64_t)e->G * i / n_threads);
    jobs[i].g1 = (int)((int64_t)e->G * (i + 1) / n_threads);
    pthread_create(&th[i], NULL, mt_worker, &jobs[i]);
  }
  for (int i = 0; i < n_threads; i++) { pthread_join(th[i], NULL); add_counters(e, &jobs[i].c); }
  e->counters[RG_C_GROUP_TICKS] += (uint64_t)e->G * (uint64_t)n_ticks;
  e->now += n_ticks;
  free(jobs); free(th);
  return RG_OK;
}

/* ---------------- verbs -------------------------------------------------- */
#define CHECK_GP(e, g, p) do { if (!(e)) return RG_ERR_ARG; if ((g) < 0 || (g) >= (e)->G || (p) < 0 || (p) >= (e)->P) return RG_ERR_RANGE; } while (0)

int ro_start(ro_engine *e, int32_t g, int32_t p, uint64_t cmd, int32_t *index, int32_t *term, int32_t *leader) {
  CHECK_GP(e, g, p);
  ctr_t c; memset(&c, 0, sizeof c);
  int32_t i, t, l;
  start_cmd(e, &c, &e->groups[g], p, cmd, &i, &t, &l);
  add_counters(e, &c);
  if (index) *index = i;
  if (term) *term = t;
  if (leader) *leader = l;
  return RG_OK;
}
int ro_get_state(ro_engine *e, int32_t g, int32_t p, int32_t *term, int32_t *leader) {
  CHECK_GP(e, g, p);
  const peer_t *rf = &e->groups[g].peer[p];
  if (term) *term = rf->term;
  if (leader) *leader = is_leader(rf);
  return RG_OK;
}
int ro_set_connected(ro_engine *e, int32_t g, int32_t p, int32_t on) { CHECK_GP(e, g, p); e->groups[g].peer[p].connected = on ? 1 : 0; return RG_OK; }
int ro_set_side(ro_engine *e, int32_t g, int32_t p, int32_t side) { CHECK_GP(e, g, p); e->groups[g].peer[p].side = side & 1; return RG_OK; }
int ro_crash(ro_engine *e, int32_t g, int32_t p) { CHECK_GP(e, g, p); crash1(&e->groups[g], p); return RG_OK; }
int ro_kill(ro_engine *e, int32_t g, int32_t p) { CHECK_GP(e, g, p); kill1(&e->groups[g], p); return RG_OK; }
int ro_restart(ro_engine *e, int32_t g, int32_t p) {
  CHECK_GP(e, g, p);
  ctr_t c; memset(&c, 0, sizeof c);
  start1(e, &c, &e->groups[g], p, (int32_t)e->now * RG_TICK_MS);
  add_counters(e, &c);
  return RG_OK;
}
int ro_set_schedule(ro_engine *e, const rg_sched_op *ops, int64_t n) {
  if (!e || n < 0 || (n > 0 && !ops)) return RG_ERR_ARG;
  for (int64_t i = 0; i < n; i++)
    if (ops[i].op < RG_OP_CONNECT || ops[i].op >= RG_OP_END || ops[i].tick < 0 || (i > 0 && ops[i].tick < ops[i - 1].tick)) return RG_ERR_ARG;
  free(e->sched);
  e->sched = NULL; e->schedN = 0;
  if (n > 0) {
    e->sched = (rg_sched_op *)malloc(sizeof(rg_sched_op) * (size_t)n);
    if (!e->sched) return RG_ERR_NOMEM;
    memcpy(e->sched, ops, sizeof(rg_sched_op) * (size_t)n);
    e->schedN = (int)n;
  }
  int first = 0;
  while (first < n && ops[first].tick < e->now) first++;
  for (int g = 0; g < e->G; g++) e->groups[g].schedPos = first;
  return RG_OK;
}
int ro_set_unreliable_group(ro_engine *e, int32_t g, int32_t a, int32_t b, int32_t d) {
  if (!e || a < 0 || b < 0 || d < 0 || a > 1000 || b > 1000 || d > 1000) return RG_ERR_ARG;
  if (g < 0 || g >= e->G) return RG_ERR_RANGE;
  group_t *gr = &e->groups[g];
  gr->netOwn = 1; gr->reqDrop = a; gr->replyDrop = b; gr->replyDelay = d;
  return RG_OK;
}
int ro_reset_groups(ro_engine *e, const int32_t *groups, int64_t n) {
  if (!e || n < 0 || (n > 0 && !groups)) return RG_ERR_ARG;
  for (int64_t i = 0; i < n; i++) if (groups[i] < 0 || groups[i] >= e->G) return RG_ERR_RANGE;
  for (int64_t i = 0; i < n; i++) init_group(e, &e->groups[groups[i]], e->cfg.group_base + groups[i], (int32_t)e->now);
  return RG_OK;
}
int ro_set_unreliable(ro_engine *e, int32_t a, int32_t b, int32_t d) {
  if (!e || a < 0 || b < 0 || d < 0 || a > 1000 || b > 1000 || d > 1000) return RG_ERR_ARG;
  e->cfg.req_drop_permille = a; e->cfg.reply_drop_permille = b; e->cfg.reply_delay_permille = d;
  return RG_OK;
}

/* the reader's end of applyCh: the messages sent since the last call, oldest first */
int ro_drain_applied(ro_engine *e, int32_t g, int32_t p, rg_apply_msg *buf, int64_t cap, int64_t *n_out) {
  CHECK_GP(e, g, p);
  if (!n_out || cap < 0 || (cap > 0 && !buf)) return RG_ERR_ARG;
  *n_out = 0;
  peer_t *rf = &e->groups[g].peer[p];
  if (!rf->applyRing) return RG_ERR_STATE;
  const uint32_t rc = (uint32_t)e->cfg.apply_ring_cap;
  int status = RG_OK;
  if (rf->applyHead - rf->applyDrained > rc) { rf->applyDrained = rf->applyHead - rc; status = RG_ERR_OVERRUN; }
  int64_t n = (int64_t)(rf->applyHead - rf->applyDrained);
  if (n > cap) n = cap;
  for (int64_t i = 0; i < n; i++) buf[i] = rf->applyRing[(rf->applyDrained + (uint32_t)i) % rc];
  rf->applyDrained += (uint32_t)n;
  *n_out = n;
  return status;
}

int ro_rpc_request_vote(ro_engine *e, int32_t g, int32_t p, const rg_vote_args *a, rg_vote_reply *r) {
  CHECK_GP(e, g, p);
  if (!a || !r) return RG_ERR_ARG;
  ctr_t c; memset(&c, 0, sizeof c);
  request_vote(e, &c, &e->groups[g].peer[p], (int32_t)e->now * RG_TICK_MS, a, r);
  add_counters(e, &c);
  return RG_OK;
}
int ro_rpc_append_entries(ro_engine *e, int32_t g, int32_t p, const rg_append_args *a,
                          const int32_t *et, const uint64_t *ec, rg_append_reply *r) {
  CHECK_GP(e, g, p);
  if (!a || !r) return RG_ERR_ARG;
  if (a->prev_log_index < 0 || a->n_entries < 0 || (int64_t)a->prev_log_index + 1 + a->n_entries > e->L) return RG_ERR_RANGE;
  if (a->n_entries > 0 && (!et || !ec)) return RG_ERR_ARG;
  ctr_t c; memset(&c, 0, sizeof c);
  append_entries(e, &c, &e->groups[g], &e->groups[g].peer[p], (int32_t)e->now * RG_TICK_MS, a,
                 et ? et + a->entry_off : NULL, ec ? ec + a->entry_off : NULL, r);
  add_counters(e, &c);
  return RG_OK;
}

int32_t ro_leader_backoff(ro_engine *e, int32_t g, int32_t leader, int32_t follower, int32_t ct, int32_t cf) {
  return leader_backoff(&e->groups[g].peer[leader], follower, ct, cf);
}
int32_t ro_commit_rule(ro_engine *e, int32_t g, int32_t leader) {
  ctr_t c; memset(&c, 0, sizeof c);
  int r = commit_rule(e, &c, &e->groups[g].peer[leader], leader);
  add_counters(e, &c);
  return r;
}

/* ---------------- bulk state access -------------------------------------- */
static int32_t *peer_i32(peer_t *rf, int field) {
  switch (field) {
  case RG_F_TERM: return &rf->term;       case RG_F_VOTE: return &rf->vote;       case RG_F_STATE: return &rf->state;
  case RG_F_COMMIT: return &rf->commit;   case RG_F_APPLIED: return &rf->applied; case RG_F_LOGLEN: return &rf->logLen;
  case RG_F_TIMER: return &rf->timer;     case RG_F_TIMEOUT: return &rf->timeout; case RG_F_HBTIMER: return &rf->hbTimer;
  case RG_F_STOP: return &rf->stop;       case RG_F_CNT: return &rf->cnt;         case RG_F_DEAD: return &rf->dead;
  case RG_F_RESTART: return &rf->restart; case RG_F_FIRST: return &rf->first;     case RG_F_MONALIVE: return &rf->monAlive;
  case RG_F_ELNEXT: return &rf->elNext;   case RG_F_MONNEXT: return &rf->monNext;
  case RG_F_RNGCTR: return (int32_t *)&rf->rngCtr; case RG_F_ROUNDSEQ: return (int32_t *)&rf->roundSeq;
  case RG_F_SNAPHAS: return &rf->snapHas; case RG_F_SNAPTERM: return &rf->snapTerm; case RG_F_SNAPVOTE: return &rf->snapVote;
  case RG_F_SNAPLEN: return &rf->snapLen; case RG_F_CONNECTED: return &rf->connected; case RG_F_SIDE: return &rf->side;
  default: return NULL;
  }
}
static int32_t *group_i32(group_t *g, int field) {
  switch (field) {
  case RG_F_STARTSEQ: return &g->startSeq; case RG_F_PARTNEXT: return &g->partNext; case RG_F_PARTEPOCH: return &g->partEpoch;
  case RG_F_RECMAX: return &g->recMax;     case RG_F_GFLAGS: return (int32_t *)&g->flags;
  case RG_F_SCHEDPOS: return &g->schedPos; case RG_F_SCHEDREGS: return &g->schedRegs;
  default: return NULL;
  }
}

static int64_t field_bytes(const ro_engine *e, int field) {
  int64_t G = e->G, P = e->P, L = e->L;
  if (field >= 0 && field < RG_F_PEER_I32_END) return G * P * 4;
  switch (field) {
  case RG_F_APPLYHASH: return G * P * 8;
  case RG_F_NEXTINDEX: case RG_F_MATCHINDEX: return G * P * P * 4;
  case RG_F_LOGTERM: case RG_F_SNAPLOGTERM: return G * P * L * 4;
  case RG_F_LOGCMD: case RG_F_SNAPLOGCMD: return G * P * L * 8;
  case RG_F_STARTSEQ: case RG_F_PARTNEXT: case RG_F_PARTEPOCH: case RG_F_RECMAX: case RG_F_GFLAGS:
  case RG_F_SCHEDPOS: case RG_F_SCHEDREGS: return G * 4;
  case RG_F_APPLYREC: return G * L * 8;
  default: return -1;
  }
}

static int rw_state(ro_engine *e, int32_t field, void *buf, int64_t bytes, int wr) {
  if (!e || !buf) return RG_ERR_ARG;
  int64_t need = field_bytes(e, field);
  if (need < 0 || bytes != need) return RG_ERR_ARG;
  const int G = e->G, P = e->P, L = e->L;
  for (int g = 0; g < G; g++) {
    group_t *gr = &e->groups[g];
    if (field >= RG_F_STARTSEQ && field <= RG_F_SCHEDREGS) {
      int32_t *w = group_i32(gr, field), *b = (int32_t *)buf + g;
      if (wr) *w = *b; else *b = *w;
      continue;
    }
    if (field == RG_F_APPLYREC) {
      uint64_t *b = (uint64_t *)buf + (int64_t)g * L;
      if (wr) memcpy(gr->applyRec, b, sizeof(uint64_t) * (size_t)L); else memcpy(b, gr->applyRec, sizeof(uint64_t) * (size_t)L);
      continue;
    }
    for (int p = 0; p < P; p++) {
      peer_t *rf = &gr->peer[p];
      int64_t gp = (int64_t)g * P + p;
      if (field < RG_F_PEER_I32_END) {
        int32_t *w = peer_i32(rf, field), *b = (int32_t *)buf + gp;
        if (wr) *w = *b; else *b = *w;
      } else if (field == RG_F_APPLYHASH) {
        uint64_t *b = (uint64_t *)buf + gp;
        if (wr) rf->applyHash = *b; else *b = rf->applyHash;
      } else if (field == RG_F_NEXTINDEX || field == RG_F_MATCHINDEX) {
        int32_t *w = field == RG_F_NEXTINDEX ? rf->next : rf->match, *b = (int32_t *)buf + gp * P;
        if (wr) memcpy(w, b, sizeof(int32_t) * (size_t)P); else memcpy(b, w, sizeof(int32_t) * (size_t)P);
      } else if (field == RG_F_LOGTERM || field == RG_F_SNAPLOGTERM) {
        int32_t *w = field == RG_F_LOGTERM ? rf->logTerm : rf->snapTerm_, *b = (int32_t *)buf + gp * L;
        if (wr) memcpy(w, b, sizeof(int32_t) * (size_t)L); else memcpy(b, w, sizeof(int32_t) * (size_t)L);
      } else {
        uint64_t *w = field == RG_F_LOGCMD ? rf->logCmd : rf->snapCmd_, *b = (uint64_t *)buf + gp * L;
        if (wr) memcpy(w, b, sizeof(uint64_t) * (size_t)L); else memcpy(b, w, sizeof(uint64_t) * (size_t)L);
      }
    }
  }
  if (wr && (field == RG_F_LOGTERM || field == RG_F_LOGCMD || field == RG_F_SNAPLOGTERM || field == RG_F_SNAPLOGCMD))
    for (int g = 0; g < G; g++) for (int p = 0; p < P; p++) e->groups[g].peer[p].snapSync = 0;   /* arrays no longer known equal */
  return RG_OK;
}
int ro_read_state(ro_engine *e, int32_t field, void *dst, int64_t bytes) { return rw_state(e, field, dst, bytes, 0); }
int ro_write_state(ro_engine *e, int32_t field, const void *src, int64_t bytes) { return rw_state(e, field, (void *)src, bytes, 1); }

/* order-free digest: sum over every Raft-visible word of H(position, value) */
static inline uint64_t elem_hash(uint32_t peer, uint32_t field, uint32_t idx, uint64_t v) {
  return sm64(sm64(((uint64_t)peer << 48) ^ ((uint64_t)field << 32) ^ idx) ^ v);
}
int ro_state_hash(ro_engine *e, uint64_t *out, int64_t n) {
  if (!e || !out || n != e->G) return RG_ERR_ARG;
  for (int g = 0; g < e->G; g++) {
    group_t *gr = &e->groups[g];
    uint64_t s = 0;
    for (int p = 0; p < e->P; p++) {
      peer_t *rf = &gr->peer[p];
      for (int f = 0; f < RG_F_PEER_I32_END; f++) s += elem_hash((uint32_t)p, (uint32_t)f, 0, (uint64_t)(uint32_t)*peer_i32(rf, f));
      s += elem_hash((uint32_t)p, RG_F_APPLYHASH, 0, rf->applyHash);
      for (int i = 0; i < e->P; i++) {
        s += elem_hash((uint32_t)p, RG_F_NEXTINDEX, (uint32_t)i, (uint64_t)(uint32_t)rf->next[i]);
        s += elem_hash((uint32_t)p, RG_F_MATCHINDEX, (uint32_t)i, (uint64_t)(uint32_t)rf->match[i]);
      }
      for (int i = 0; i < rf->logLen && i < e->L; i++) {
        s += elem_hash((uint32_t)p, RG_F_LOGTERM, (uint32_t)i, (uint64_t)(uint32_t)rf->logTerm[i]);
        s += elem_hash((uint32_t)p, RG_F_LOGCMD, (uint32_t)i, rf->logCmd[i]);
      }
      for (int i = 0; i < rf->snapLen && i < e->L; i++) {
        s += elem_hash((uint32_t)p, RG_F_SNAPLOGTERM, (uint32_t)i, (uint64_t)(uint32_t)rf->snapTerm_[i]);
        s += elem_hash((uint32_t)p, RG_F_SNAPLOGCMD, (uint32_t)i, rf->snapCmd_[i]);
      }
    }
    for (int f = RG_F_STARTSEQ; f <= RG_F_GFLAGS; f++) s += elem_hash(255, (uint32_t)f, 0, (uint64_t)(uint32_t)*group_i32(gr, f));
    for (int i = 1; i <= gr->recMax && i < e->L; i++) s += elem_hash(255, RG_F_APPLYREC, (uint32_t)i, gr->applyRec[i]);
    out[g] = sm64(s ^ (uint64_t)(e->cfg.group_base + g));
  }
  return RG_OK;
}

int ro_counters(ro_engine *e, uint64_t *out) {
  if (!e || !out) return RG_ERR_ARG;
  memcpy(out, e->counters, sizeof e->counters);
  return RG_OK;
}

/* cfg.bytesTotal() under the fixed-width size model of include/raftgpu.h (labrpc.go:153-154,286-290) */
int ro_rpc_bytes(ro_engine *e, int64_t command_bytes, uint64_t *total) {
  if (!e || !total || command_bytes < 0) return RG_ERR_ARG;
  if (!e->cfg.account_bytes) return RG_ERR_STATE;
  const uint64_t *c = e->counters;
  *total = c[RG_C_RPC_FIXED_BYTES] + (4 + (uint64_t)command_bytes) * c[RG_C_AE_ENTRIES_SENT];
  return RG_OK;
}

/* Full scan.  Log matching: for two peers holding index i with the same term,
 * the commands at i and the terms at i-1 must agree (by induction the whole
 * prefix).  Leader completeness: every index some peer has applied must be in
 * every live, self-declared leader's log with the applied command. */
int ro_safety_check(ro_engine *e) {
  if (!e) return RG_ERR_ARG;
  for (int g = 0; g < e->G; g++) {
    group_t *gr = &e->groups[g];
    for (int a = 0; a < e->P; a++) for (int b = a + 1; b < e->P; b++) {
      peer_t *x = &gr->peer[a], *y = &gr->peer[b];
      int n = x->logLen < y->logLen ? x->logLen : y->logLen;
      for (int i = 1; i < n; i++)
        if (x->logTerm[i] == y->logTerm[i] && (x->logCmd[i] != y->logCmd[i] || x->logTerm[i - 1] != y->logTerm[i - 1]))
          gr->flags |= RG_FLAG_LOG_MATCHING;
    }
    for (int a = 0; a < e->P; a++) {
      peer_t *x = &gr->peer[a];
      if (!is_leader(x)) continue;
      for (int i = 1; i <= gr->recMax; i++)
        if (i >= x->logLen || x->logCmd[i] != gr->applyRec[i]) gr->flags |= RG_FLAG_LEADER_COMPLETENESS;
    }
  }
  return RG_OK;
}
int ro_safety_flags(ro_engine *e, uint32_t *out) {
  if (!e || !out) return RG_ERR_ARG;
  uint32_t o = 0, n = 0;
  for (int g = 0; g < e->G; g++) { o |= e->groups[g].flags; n += e->groups[g].flags != 0; }
  out[0] = o; out[1] = n;
  return RG_OK;
}

/* rg_safety_summary's words: groups per flag bit, flagged groups, and how much of the run's work the flagged
 * groups did (rounds their peers sent; highest index their testers recorded). */
int ro_safety_summary(ro_engine *e, uint64_t *out) {
  if (!e || !out) return RG_ERR_ARG;
  memset(out, 0, sizeof(uint64_t) * RG_N_SUMMARY);
  for (int g = 0; g < e->G; g++) {
    const group_t *gr = &e->groups[g];
    uint64_t rounds = 0;
    for (int p = 0; p < e->P; p++) rounds += gr->peer[p].roundSeq;
    for (int b = 0; b < RG_N_FLAG_BITS; b++) if ((gr->flags >> b) & 1u) out[RG_S_BIT_GROUPS + b]++;
    out[RG_S_ROUNDS_ALL] += rounds;
    out[RG_S_RECORDED_ALL] += (uint64_t)gr->recMax;
    if (gr->flags) { out[RG_S_FLAGGED_GROUPS]++; out[RG_S_ROUNDS_FLAGGED] += rounds; out[RG_S_RECORDED_FLAGGED] += (uint64_t)gr->recMax; }
  }
  return RG_OK;
}

/* Persister blob, layout in include/raftgpu.h */
int ro_snapshot_export(ro_engine *e, int32_t g, int32_t p, void *buf, int64_t cap, int64_t *size_out) {
  CHECK_GP(e, g, p);
  peer_t *rf = &e->groups[g].peer[p];
  int32_t n = rf->snapHas ? rf->snapLen : 0;
  int64_t size = rf->snapHas ? 16 + 16 * (int64_t)n : 0;
  if (size_out) *size_out = size;
  if (!buf) return RG_OK;
  if (cap < size) return RG_ERR_ARG;
  if (!rf->snapHas) return RG_OK;
  int32_t *h = (int32_t *)buf;
  h[0] = rf->snapTerm; h[1] = rf->snapVote; h[2] = n; h[3] = 0;
  for (int i = 0; i < n; i++) {
    int32_t *t = (int32_t *)((char *)buf + 16 + 16 * (int64_t)i);
    t[0] = rf->snapTerm_[i]; t[1] = 0;
    memcpy(t + 2, &rf->snapCmd_[i], 8);
  }
  return RG_OK;
}
int ro_snapshot_import(ro_engine *e, int32_t g, int32_t p, const void *buf, int64_t size) {
  CHECK_GP(e, g, p);
  peer_t *rf = &e->groups[g].peer[p];
  if (size == 0) { rf->snapHas = 0; rf->snapLen = 0; return RG_OK; }
  if (!buf || size < 16) return RG_ERR_ARG;
  const int32_t *h = (const int32_t *)buf;
  int32_t n = h[2];
  if (n < 0 || n > e->L || size != 16 + 16 * (int64_t)n) return RG_ERR_ARG;
  rf->snapTerm = h[0]; rf->snapVote = h[1]; rf->snapLen = n; rf->snapHas = 1;
  for (int i = 0; i < n; i++) {
    const int32_t *t = (const int32_t *)((const char *)buf + 16 + 16 * (int64_t)i);
    rf->snapTerm_[i] = t[0];
    memcpy(&rf->snapCmd_[i], t + 2, 8);
  }
  rf->snapSync = 0;
  return RG_OK;
}
