/* The C ABI used from plain C, the way a cgo or C++ host would: the reference's first agreement test
 * (src/raft/test_test.go:92-113: three peers, three commands, indices 1, 2, 3) on one group of a 64-group engine.
 * Built by __graft_entry__.build(); run by tests/test_gpu_api.py on a GPU. */
#include <stdio.h>
#include <string.h>
#include "raftgpu.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != RG_OK) { printf("FAIL %s: %s\n", #x, rg_strerror(rc_)); return 1; } } while (0)

int main(void) {
  if (rg_device_count() < 1) { printf("FAIL no gfx950 device\n"); return 2; }
  rg_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_groups = 64; cfg.n_peers = 3; cfg.log_cap = 64; cfg.seed = 1; cfg.apply_ring_cap = 64;   /* applyCh is read below */
  rg_engine *e = NULL;
  CHECK(rg_create(&cfg, &e));
  { int rc = rg_specialize(e);                             /* a kernel compiled for this configuration, where hiprtc is at hand */
    if (rc != RG_OK && rc != RG_ERR_STATE) { printf("FAIL rg_specialize: %s\n", rg_strerror(rc)); return 1; } }
  CHECK(rg_step(e, 100));                                  /* one second: an election */
  for (int k = 1; k <= 3; k++) {
    int32_t leader = -1, term = 0, is_leader = 0;
    for (int32_t p = 0; p < 3; p++) {
      CHECK(rg_get_state(e, 7, p, &term, &is_leader));
      if (is_leader) leader = p;
    }
    if (leader < 0) { printf("FAIL no leader\n"); return 1; }
    int32_t index = 0;
    CHECK(rg_start(e, 7, leader, (uint64_t)(100 * k), &index, &term, &is_leader));
    if (!is_leader || index != k) { printf("FAIL Start returned index %d, want %d\n", index, k); return 1; }
    CHECK(rg_step(e, 30));
    for (int32_t p = 0; p < 3; p++) {                      /* every peer delivers (k, 100k) next */
      rg_apply_msg m[4];
      int64_t n = 0;
      CHECK(rg_drain_applied(e, 7, p, m, 4, &n));
      if (n != 1 || m[0].command_index != k || m[0].command != (uint64_t)(100 * k)) { printf("FAIL peer %d applied %lld msgs\n", p, (long long)n); return 1; }
    }
  }
  uint32_t flags[2] = {1, 1};
  CHECK(rg_safety_check(e));
  CHECK(rg_safety_flags(e, flags));
  if (flags[0] || flags[1]) { printf("FAIL safety flags %u\n", flags[0]); return 1; }
  if (rg_start(e, 99, 0, 1, NULL, NULL, NULL) == RG_OK) { printf("FAIL out-of-range group accepted\n"); return 1; }
  CHECK(rg_destroy(e));
  printf("OK\n");
  return 0;
}
