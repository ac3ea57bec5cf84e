// Drives the linearizability checker (raft_amd/csrc/lincheck.cpp) under ASan/UBSan: random concurrent key/value
// histories, legal and corrupted ones.  Run by tests/test_oracle_sanitize.py.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/lincheck.h"

int main() {
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  int legal = 0, illegal = 0;
  for (int round = 0; round < 200; round++) {
    std::vector<std::string> keep;
    keep.reserve(4096);
    std::vector<lc_kv_op> ops;
    std::string state[3];
    long long t = 0;
    const int n = 1 + rnd() % 40;
    for (int i = 0; i < n; i++) {             // a sequential history (always linearizable), calls overlapping their successor's at random
      lc_kv_op o;
      o.client_id = rnd() % 5; o.op = rnd() % 3;
      const int k = rnd() % 3;
      keep.push_back(std::to_string(k)); o.key = keep.back().c_str();
      keep.push_back("v" + std::to_string(rnd() % 9)); o.value = keep.back().c_str();
      if (o.op == LC_KV_PUT) state[k] = o.value;
      if (o.op == LC_KV_APPEND) state[k] += o.value;
      keep.push_back(o.op == LC_KV_GET ? state[k] : std::string()); o.output = keep.back().c_str();
      o.call = t; t += 1 + rnd() % 3; o.ret = t + rnd() % 4;
      ops.push_back(o);
    }
    int32_t res = -1;
    if (lc_check_kv_operations(ops.data(), (int64_t)ops.size(), 2000, &res) != 0 || res != LC_OK) { fprintf(stderr, "legal history rejected\n"); return 1; }
    legal++;
    for (auto &o : ops)
      if (o.op == LC_KV_GET) { keep.push_back(std::string(o.output) + "!"); o.output = keep.back().c_str(); illegal++;
        if (lc_check_kv_operations(ops.data(), (int64_t)ops.size(), 2000, &res) != 0 || res != LC_ILLEGAL) { fprintf(stderr, "corrupted history accepted\n"); return 1; }
        break; }
  }
  uint64_t w[3] = {1, 2, 3};
  if (lc_bitset_hash(w, 3) == lc_bitset_hash(w, 2)) return 1;
  int32_t res = -1;
  if (lc_check_kv_operations(nullptr, 0, 10, &res) != 0 || res != LC_OK) return 1;
  printf("sanitize OK: %d legal, %d corrupted\n", legal, illegal);
  return 0;
}
