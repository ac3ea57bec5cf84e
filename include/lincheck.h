/* lincheck.h -- C ABI of the host-side linearizability checker.
 *
 * CPU code by nature (a backtracking search over one client history): it replaces the reference's
 * `porcupine.CheckOperations` / `CheckOperationsTimeout` / `CheckEvents` (src/porcupine/porcupine.go:5-39,
 * algorithm in src/porcupine/checker.go:43-55,116-248, bitset in src/porcupine/bitset.go) with the
 * key/value model of src/models/kv.go:17-54 built in.  Histories come from the apply streams of
 * sampled groups of the engine (raft_amd/kvservice.py).
 */
#ifndef LINCHECK_H
#define LINCHECK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* porcupine.CheckResult, src/porcupine/model.go:71-77 */
#define LC_ILLEGAL 0
#define LC_OK 1
#define LC_UNKNOWN 2 /* timed out */

/* models.KvInput.Op, src/models/kv.go:7-11 */
#define LC_KV_GET 0
#define LC_KV_PUT 1
#define LC_KV_APPEND 2

/* porcupine.Operation with Input = models.KvInput and Output = models.KvOutput
 * (src/porcupine/model.go:5-11, src/models/kv.go:7-15).  Strings are NUL-terminated; NULL reads as "". */
typedef struct lc_kv_op {
  int32_t client_id;
  int32_t op;         /* LC_KV_* */
  const char *key;
  const char *value;  /* input value (put, append) */
  const char *output; /* output value (get) */
  int64_t call;       /* invocation time */
  int64_t ret;        /* response time */
} lc_kv_op;

/* porcupine.Event (src/porcupine/model.go:20-25): a call carries (op, key, value), the return with the same
 * id carries the output in `value`. */
typedef struct lc_kv_event {
  int32_t client_id;
  int32_t is_return;
  int32_t id;
  int32_t op;
  const char *key;
  const char *value;
} lc_kv_event;

/* CheckOperationsTimeout(models.KvModel, history, timeout), src/porcupine/porcupine.go:13-18: the history is
 * partitioned by key (kv.go:18-34), every partition searched (checker.go:174-248).  timeout_ms <= 0 waits for
 * ever (porcupine.go:5-8).  *result = LC_OK / LC_ILLEGAL / LC_UNKNOWN.  Returns 0, or -1 for bad arguments
 * (NULL pointers, an operation whose return precedes its call, an unknown op). */
int lc_check_kv_operations(const lc_kv_op *ops, int64_t n, int64_t timeout_ms, int32_t *result);

/* CheckEventsTimeout(models.KvModel, history, timeout), src/porcupine/porcupine.go:29-32: events in history
 * order, one partition (KvModel has no PartitionEvent: model.go:55-57), ids renumbered (checker.go:87-101).
 * Returns -1 also for a return without a call or an id with other than one call and one return. */
int lc_check_kv_events(const lc_kv_event *events, int64_t n, int64_t timeout_ms, int32_t *result);

/* bitset.hash() of src/porcupine/bitset.go:54-60, exported for known-answer tests. */
uint64_t lc_bitset_hash(const uint64_t *words, int64_t n_words);

/* Number of (state, linearized-set) pairs the last search on this thread put in its cache, summed over the
 * partitions (checker.go:195-197): a cost figure for tests. */
int64_t lc_last_cache_size(void);

#ifdef __cplusplus
}
#endif
#endif
