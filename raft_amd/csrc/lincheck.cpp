// lincheck.cpp -- linearizability checker for client histories (host code, C ABI in include/lincheck.h).
//
// Same search as the reference's porcupine package (src/porcupine/checker.go:116-248): walk the
// time-ordered list of call/return entries; at a call, try to linearize it next (model step + a cache of
// (linearized set, state) pairs already explored); at a return whose call is not linearized yet,
// backtrack.  The model is the single-key string register of src/models/kv.go:35-54, histories are
// split per key (kv.go:18-34).  Entries live in index-linked arrays instead of heap nodes.
#include "../../include/lincheck.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct KvInput { int op; std::string value; };
struct Entry {              // checker.go:16-22
  bool isReturn;
  int id;
  int64_t time;
  int opIndex;              // which operation (call: its input; return: its output)
};
struct Op { KvInput in; std::string out; };

// bitset.go:5-72
struct Bitset {
  std::vector<uint64_t> w;
  explicit Bitset(size_t bits = 0) : w((bits + 63) / 64, 0) {}
  void set(size_t pos) { w[pos / 64] |= (uint64_t)1 << (pos % 64); }
  void clear(size_t pos) { w[pos / 64] &= ~((uint64_t)1 << (pos % 64)); }
  uint64_t hash() const { return lc_bitset_hash(w.data(), (int64_t)w.size()); }
  bool operator==(const Bitset &o) const { return w == o.w; }
};

struct Node {               // checker.go:57-63; -1 = nil
  int match;                // the return node of a call; -1 on a return node
  int id;
  int opIndex;
  int next, prev;
};

struct CacheEntry { Bitset linearized; std::string state; };

thread_local int64_t g_cacheSize = 0;

// KvModel.Step, kv.go:40-54
inline bool kv_step(const std::string &state, const Op &op, std::string &newState) {
  if (op.in.op == LC_KV_GET) { newState = state; return op.out == state; }
  if (op.in.op == LC_KV_PUT) { newState = op.in.value; return true; }
  newState = state + op.in.value;
  return true;
}

using Clock = std::chrono::steady_clock;

// checkSingle, checker.go:174-248 (without the partial-linearization bookkeeping of the visualizer).
// Returns 1 linearizable, 0 not, 2 deadline passed.
int check_single(const std::vector<Entry> &history, const std::vector<Op> &ops, bool haveDeadline, Clock::time_point deadline) {
  // makeLinkedEntries, checker.go:116-133: nodes in history order; node 0 is the extra head (checker.go:184)
  const int m = (int)history.size();
  std::vector<Node> nd((size_t)m + 1);
  nd[0] = Node{-1, -1, -1, m > 0 ? 1 : -1, -1};
  {
    std::unordered_map<int, int> retOf;
    for (int i = m - 1; i >= 0; i--) {
      const Entry &e = history[(size_t)i];
      Node &x = nd[(size_t)i + 1];
      x.id = e.id; x.opIndex = e.opIndex; x.prev = i; x.next = i + 2 <= m ? i + 2 : -1;
      if (e.isReturn) { x.match = -1; retOf[e.id] = i + 1; }
      else x.match = retOf.count(e.id) ? retOf[e.id] : -1;
    }
  }
  const size_t n = (size_t)m / 2;
  Bitset linearized(n);
  std::unordered_map<uint64_t, std::vector<CacheEntry>> cache;
  struct Call { int entry; std::string state; };
  std::vector<Call> calls;
  std::string state;                       // KvModel.Init, kv.go:35-39
  std::string newState;
  int entry = nd[0].next;
  unsigned spin = 0;
  while (nd[0].next != -1) {
    if (haveDeadline && (++spin & 1023u) == 0 && Clock::now() > deadline) return 2;
    Node &e = nd[(size_t)entry];
    if (e.match != -1) {
      bool advanced = false;
      if (kv_step(state, ops[(size_t)e.opIndex], newState)) {
        Bitset nl = linearized;
        nl.set((size_t)e.id);
        const uint64_t h = nl.hash();
        std::vector<CacheEntry> &bucket = cache[h];
        bool seen = false;
        for (const CacheEntry &c : bucket)
          if (c.linearized == nl && c.state == newState) { seen = true; break; }
        if (!seen) {
          bucket.push_back(CacheEntry{nl, newState});
          g_cacheSize++;
          calls.push_back(Call{entry, state});
          state = newState;
          linearized.set((size_t)e.id);
          // lift, checker.go:154-162: unlink the call and its return
          nd[(size_t)e.prev].next = e.next;
          nd[(size_t)e.next].prev = e.prev;
          Node &r = nd[(size_t)e.match];
          nd[(size_t)r.prev].next = r.next;
          if (r.next != -1) nd[(size_t)r.next].prev = r.prev;
          entry = nd[0].next;
          advanced = true;
        }
      }
      if (!advanced) entry = e.next;
    } else {
      if (calls.empty()) return 0;
      Call top = calls.back();
      calls.pop_back();
      entry = top.entry;
      state = top.state;
      Node &c = nd[(size_t)entry];
      linearized.clear((size_t)c.id);
      // unlift, checker.go:164-172
      Node &r = nd[(size_t)c.match];
      nd[(size_t)r.prev].next = c.match;
      if (r.next != -1) nd[(size_t)r.next].prev = c.match;
      nd[(size_t)c.prev].next = entry;
      nd[(size_t)c.next].prev = entry;
      entry = c.next;
    }
  }
  return 1;
}

int run_partitions(const std::vector<std::vector<Entry>> &parts, const std::vector<Op> &ops, int64_t timeout_ms, int32_t *result) {
  // checkParallel, checker.go:269-348: the reference searches the partitions in goroutines and stops at the
  // first illegal one; here they run one after the other under one deadline.
  const bool haveDeadline = timeout_ms > 0;
  const Clock::time_point deadline = Clock::now() + std::chrono::milliseconds(haveDeadline ? timeout_ms : 0);
  g_cacheSize = 0;
  bool timedOut = false;
  for (const auto &p : parts) {
    const int r = check_single(p, ops, haveDeadline, deadline);
    if (r == 0) { *result = LC_ILLEGAL; return 0; }
    if (r == 2) { timedOut = true; break; }
  }
  *result = timedOut ? LC_UNKNOWN : LC_OK;
  return 0;
}

inline std::string str(const char *s) { return s ? std::string(s) : std::string(); }

}  // namespace

extern "C" uint64_t lc_bitset_hash(const uint64_t *words, int64_t n_words) {
  uint64_t pop = 0, x = 0;
  for (int64_t i = 0; i < n_words; i++) { pop += (uint64_t)__builtin_popcountll(words[i]); x ^= words[i]; }
  return pop ^ x;
}

extern "C" int64_t lc_last_cache_size(void) { return g_cacheSize; }

extern "C" int lc_check_kv_operations(const lc_kv_op *ops, int64_t n, int64_t timeout_ms, int32_t *result) {
  if (!result || n < 0 || (n > 0 && !ops)) return -1;
  std::vector<Op> table((size_t)n);
  std::map<std::string, std::vector<int>> byKey;           // sorted by key, kv.go:24-33
  for (int64_t i = 0; i < n; i++) {
    if (ops[i].op < LC_KV_GET || ops[i].op > LC_KV_APPEND || ops[i].ret < ops[i].call) return -1;
    table[(size_t)i] = Op{KvInput{ops[i].op, str(ops[i].value)}, str(ops[i].output)};
    byKey[str(ops[i].key)].push_back((int)i);
  }
  std::vector<std::vector<Entry>> parts;
  for (const auto &kv : byKey) {
    // makeEntries, checker.go:43-55.  sort.Sort leaves the order of equal times open; here ties keep the
    // order call-before-return of the same operation and history order otherwise.
    std::vector<Entry> es;
    int id = 0;
    for (int oi : kv.second) {
      es.push_back(Entry{false, id, ops[oi].call, oi});
      es.push_back(Entry{true, id, ops[oi].ret, oi});
      id++;
    }
    std::stable_sort(es.begin(), es.end(), [](const Entry &a, const Entry &b) { return a.time < b.time; });
    parts.push_back(std::move(es));
  }
  return run_partitions(parts, table, timeout_ms, result);
}

extern "C" int lc_check_kv_events(const lc_kv_event *events, int64_t n, int64_t timeout_ms, int32_t *result) {
  if (!result || n < 0 || (n > 0 && !events)) return -1;
  // renumber + convertEntries, checker.go:87-114: ids in order of first appearance, position as time
  std::unordered_map<int, int> renum;
  std::vector<Op> table;
  std::vector<int> nCalls, nRets;
  std::vector<Entry> es;
  for (int64_t i = 0; i < n; i++) {
    const lc_kv_event &ev = events[i];
    auto it = renum.find(ev.id);
    int id;
    if (it == renum.end()) {
      id = (int)table.size();
      renum[ev.id] = id;
      table.push_back(Op{KvInput{LC_KV_GET, std::string()}, std::string()});
      nCalls.push_back(0); nRets.push_back(0);
    } else id = it->second;
    if (ev.is_return) {
      if (nCalls[(size_t)id] != 1) return -1;               // a return before its call
      table[(size_t)id].out = str(ev.value);
      nRets[(size_t)id]++;
    } else {
      if (ev.op < LC_KV_GET || ev.op > LC_KV_APPEND) return -1;
      table[(size_t)id].in = KvInput{ev.op, str(ev.value)};
      nCalls[(size_t)id]++;
    }
    es.push_back(Entry{ev.is_return != 0, id, i, id});
  }
  for (size_t i = 0; i < table.size(); i++)
    if (nCalls[i] != 1 || nRets[i] != 1) return -1;
  std::vector<std::vector<Entry>> parts;
  parts.push_back(std::move(es));
  return run_partitions(parts, table, timeout_ms, result);
}
