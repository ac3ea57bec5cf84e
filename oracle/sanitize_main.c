/*
 * sanitize_main.c -- drives the CPU checker (raft_oracle.c) through every entry point under AddressSanitizer and
 * UndefinedBehaviorSanitizer (host code only; make -C oracle sanitize).  TEST INFRASTRUCTURE: run by
 * tests/test_oracle_sanitize.py.  It checks nothing but that the runs are clean and self-consistent; the results
 * themselves are held to known answers and golden digests elsewhere.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raft_oracle.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #x); exit(1); } } while (0)

static void run_config(const rg_config *cfg, int ticks, int threads) {
  ro_engine *a = ro_create(cfg), *b = ro_create(cfg);
  CHECK(a && b);
  CHECK(ro_step(a, ticks / 2) == RG_OK && ro_step(a, ticks - ticks / 2) == RG_OK);
  CHECK(ro_step_mt(b, ticks, threads) == RG_OK);
  uint64_t *ha = malloc(8 * (size_t)cfg->n_groups), *hb = malloc(8 * (size_t)cfg->n_groups);
  CHECK(ro_state_hash(a, ha, cfg->n_groups) == RG_OK && ro_state_hash(b, hb, cfg->n_groups) == RG_OK);
  CHECK(memcmp(ha, hb, 8 * (size_t)cfg->n_groups) == 0);
  uint64_t ca[RG_N_COUNTERS], cb[RG_N_COUNTERS];
  CHECK(ro_counters(a, ca) == RG_OK && ro_counters(b, cb) == RG_OK && memcmp(ca, cb, sizeof ca) == 0);
  CHECK(ro_safety_check(a) == RG_OK);
  uint32_t fl[2];
  CHECK(ro_safety_flags(a, fl) == RG_OK);
  /* every field out and in again */
  static const int fields[] = {RG_F_TERM, RG_F_SIDE, RG_F_APPLYHASH, RG_F_NEXTINDEX, RG_F_MATCHINDEX, RG_F_LOGTERM, RG_F_SNAPLOGTERM,
                               RG_F_LOGCMD, RG_F_SNAPLOGCMD, RG_F_STARTSEQ, RG_F_GFLAGS, RG_F_SCHEDPOS, RG_F_SCHEDREGS, RG_F_APPLYREC};
  const int64_t G = cfg->n_groups, P = cfg->n_peers, L = cfg->log_cap;
  for (unsigned i = 0; i < sizeof fields / sizeof fields[0]; i++) {
    const int f = fields[i];
    int64_t bytes = f < RG_F_PEER_I32_END ? G * P * 4 : f == RG_F_APPLYHASH ? G * P * 8 : (f == RG_F_NEXTINDEX || f == RG_F_MATCHINDEX) ? G * P * P * 4
                  : (f == RG_F_LOGTERM || f == RG_F_SNAPLOGTERM) ? G * P * L * 4 : (f == RG_F_LOGCMD || f == RG_F_SNAPLOGCMD) ? G * P * L * 8
                  : f == RG_F_APPLYREC ? G * L * 8 : G * 4;
    void *buf = malloc((size_t)bytes);
    CHECK(ro_read_state(a, f, buf, bytes) == RG_OK);
    CHECK(ro_read_state(a, f, buf, bytes - 1) == RG_ERR_ARG);
    if (f != RG_F_LOGTERM && f != RG_F_LOGCMD && f != RG_F_SNAPLOGTERM && f != RG_F_SNAPLOGCMD) CHECK(ro_write_state(a, f, buf, bytes) == RG_OK);
    free(buf);
  }
  /* Persister blobs, restarts, verbs, handlers, applyCh */
  for (int g = 0; g < cfg->n_groups && g < 3; g++)
    for (int p = 0; p < cfg->n_peers; p++) {
      int64_t size = 0;
      CHECK(ro_snapshot_export(a, g, p, NULL, 0, &size) == RG_OK);
      void *blob = malloc((size_t)size + 1);
      CHECK(ro_snapshot_export(a, g, p, blob, size, &size) == RG_OK);
      CHECK(ro_snapshot_import(a, g, (p + 1) % cfg->n_peers, blob, size) == RG_OK);
      CHECK(ro_restart(a, g, (p + 1) % cfg->n_peers) == RG_OK);
      CHECK(ro_set_connected(a, g, (p + 1) % cfg->n_peers, 1) == RG_OK);
      free(blob);
      int32_t idx, term, lead;
      CHECK(ro_start(a, g, p, 77, &idx, &term, &lead) == RG_OK);
      CHECK(ro_get_state(a, g, p, &term, &lead) == RG_OK);
      rg_vote_args va = {term + 1, p, 0, 0};
      rg_vote_reply vr = {0, 0};
      CHECK(ro_rpc_request_vote(a, g, (p + 1) % cfg->n_peers, &va, &vr) == RG_OK);
      int32_t et[2] = {term + 1, term + 1};
      uint64_t ec[2] = {5, 6};
      rg_append_args aa = {term + 1, p, 0, 0, 1, 2, 0};
      rg_append_reply ar = {0, 0, 0, 0};
      CHECK(ro_rpc_append_entries(a, g, (p + 1) % cfg->n_peers, &aa, et, ec, &ar) == RG_OK);
      aa.prev_log_index = (int32_t)L;                      /* past the capacity: refused, nothing read */
      CHECK(ro_rpc_append_entries(a, g, p, &aa, et, ec, &ar) == RG_ERR_RANGE);
      if (cfg->apply_ring_cap > 0) {
        rg_apply_msg msgs[8];
        int64_t n = 0;
        int rc = ro_drain_applied(a, g, p, msgs, 8, &n);
        CHECK(rc == RG_OK || rc == RG_ERR_OVERRUN);
      }
    }
  CHECK(ro_crash(a, 0, 0) == RG_OK && ro_set_side(a, 0, 1, 1) == RG_OK);
  CHECK(ro_set_unreliable(a, 300, 300, 667) == RG_OK && ro_set_unreliable_group(a, 0, 0, 0, 0) == RG_OK);
  int32_t grp = cfg->n_groups - 1;
  CHECK(ro_reset_groups(a, &grp, 1) == RG_OK);
  CHECK(ro_step(a, 300) == RG_OK);
  CHECK(ro_start(a, cfg->n_groups, 0, 1, NULL, NULL, NULL) == RG_ERR_RANGE);
  ro_reset(a);
  CHECK(ro_now(a) == 0 && ro_step(a, 50) == RG_OK);
  free(ha); free(hb);
  ro_destroy(a); ro_destroy(b);
}

int main(void) {
  rg_config c;
  memset(&c, 0, sizeof c);
  c.n_groups = 24; c.n_peers = 5; c.log_cap = 96; c.seed = 2;
  c.req_drop_permille = 100; c.reply_drop_permille = 100; c.start_period = 5; c.start_max = 90; c.part_period = 100; c.part_jitter = 20;
  c.account_bytes = 1; c.apply_ring_cap = 16;
  run_config(&c, 1500, 3);                                  /* the benchmark's net: loss and rolling partitions */
  memset(&c, 0, sizeof c);
  c.n_groups = 10; c.n_peers = 7; c.log_cap = 64; c.seed = 4; c.group_base = 123456789;
  c.req_drop_permille = 50; c.reply_drop_permille = 50; c.reply_delay_permille = 300; c.start_period = 3; c.start_max = 63; c.start_mode = RG_START_ALL;
  c.churn_period = 35; c.churn_disconnect_permille = 200; c.churn_restart_permille = 500; c.churn_crash_permille = 200; c.apply_ring_cap = 200;
  run_config(&c, 2000, 4);                                  /* crashes, restarts from snapshots, full logs */
  memset(&c, 0, sizeof c);
  c.n_groups = 6; c.n_peers = 3; c.log_cap = 32; c.seed = 9;
  ro_engine *e = ro_create(&c);
  CHECK(e);
  rg_sched_op ops[] = {{20, RG_OP_PICK_LEADER, 0, 0}, {20, RG_OP_START, RG_SEL_REG(0, 0), 1}, {25, RG_OP_DISCONNECT, RG_SEL_REG(0, 1), 0},
                       {30, RG_OP_START_ANY, 0, 2}, {40, RG_OP_CRASH, RG_SEL_REG(0, 0), 0}, {41, RG_OP_START_ALL, 1, 3}, {45, RG_OP_PICK_RANDOM, 2, 7},
                       {45, RG_OP_RESTART_IF_FEWER, RG_SEL_ALL, 9}, {50, RG_OP_PICK_OTHER, 3, RG_SEL_REG(2, 0)}, {60, RG_OP_RESTART, RG_SEL_REG(3, 0), 0},
                       {70, RG_OP_CONNECT, RG_SEL_ALL, 0}};
  CHECK(ro_set_schedule(e, ops, sizeof ops / sizeof ops[0]) == RG_OK);
  CHECK(ro_step(e, 400) == RG_OK);
  ops[3].tick = 1;                                          /* not sorted any more */
  CHECK(ro_set_schedule(e, ops, sizeof ops / sizeof ops[0]) == RG_ERR_ARG);
  CHECK(ro_set_schedule(e, NULL, 0) == RG_OK);
  ro_destroy(e);
  CHECK(ro_create(NULL) == NULL);
  puts("sanitize OK");
  return 0;
}
