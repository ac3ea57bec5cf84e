// raftgpu.hip -- batched lockstep Raft engine for MI355X (gfx950): HIP kernels
// and the C ABI of include/raftgpu.h.
//
// Mapping.  One lane per (group, peer); a group's peers sit in PP = 4 or 8
// adjacent lanes of one 64-wide wavefront, so a wave owns 16 or 8 whole groups
// and never needs another wave: every cross-peer exchange is a bpermute or a
// ballot slice.  The tick kernel loads a peer's scalar words into registers
// once, advances n ticks, and stores them back; logs, snapshot arrays and the
// leader matrices stay in HBM and are touched only by the events that need them.
//
// Log layout.  Entry i of ALL peers of a group shares one row
// {int32 term[P]; uint64 cmd[P]} (64 B for P = 5), rows consecutive in i.  A
// leader replicating entry i to its followers, the followers' snapshot copies
// and the apply reads of index i therefore all fall in the same cache line.
//
// Semantics.  A tick is: control (partition reshuffle, churn, client Starts),
// actors in peer order (election loop iteration, then leader loop), apply,
// one-leader check -- the reference's statements (src/raft/raft.go) under the
// serialisation fixed by the lockstep model.  The reference runs a round's
// RPCs one after the other and stops at the first reply with a higher term
// (raft.go:482-487, 622-627).  Here all targets of a round evaluate their
// handler at once without touching state, the break point is found with a
// ballot, and only targets up to it commit their effects: the result is the
// sequential one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/raftgpu.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------- device state
struct DevPtrs {
  // per peer, int32 [Gpad*PP]
  int *term, *vote, *flags, *commit, *applied, *logLen, *timer, *timeout, *hbTimer, *elNext, *monNext;
  int *rngCtr, *roundSeq, *snapTerm, *snapVote, *snapLen, *snapSync, *lastTerm, *firstOwn, *commitTerm;
  u64 *applyHash;
  // leader matrices int32 [Gpad][PP][PP]
  int *nextIdx, *matchIdx;
  // rows [G][L][rowBytes]
  char *logRows, *snapRows;
  u64 *applyRec;                 // [G][L]
  int *startSeq, *partNext, *partEpoch, *recMax;   // [Gpad]
  unsigned *gflags;              // [Gpad]
  u64 *counters;                 // [RG_N_COUNTERS]
  unsigned *flagSummary;         // [2]
};

struct DevCfg {
  int G, P, L, rowBytes, cmdOff;
  long long groupBase;
  u64 seedKey;                   // sm64(seed)
  int reqDrop, replyDrop, replyDelay;
  int startPeriod, startFirst, startMax, startMode;
  int partPeriod, partJitter;
  int churnPeriod, churnDisc, churnRestart, churnCrash;
  int acctBytes;
};

// packed per-peer flag word
#define F_STATE_MASK 0x3u
#define F_STOP      (1u << 2)
#define F_DEAD      (1u << 3)
#define F_RESTART   (1u << 4)
#define F_FIRST     (1u << 5)
#define F_MONALIVE  (1u << 6)
#define F_SNAPHAS   (1u << 7)
#define F_CONNECTED (1u << 8)
#define F_SIDE      (1u << 9)
#define F_CNT_SHIFT 10
#define F_CNT_MASK  (7u << F_CNT_SHIFT)
#define F_CDIRTY    (1u << 13)   // leader matrices changed since the commit rule last ran
#define F_SNAPEXT   (1u << 14)   // the persisted log lives in the snapshot rows (imported blob), not in the log rows

#define STREAM_ELECTION 1ull
#define STREAM_NET      2ull
#define STREAM_CMD      3ull
#define STREAM_PART     4ull
#define STREAM_CHURN    5ull

__host__ __device__ static inline u64 sm64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ static inline u64 stream_hash(u64 key, u64 stream, unsigned a, unsigned b) {
  return sm64(key ^ (stream << 56) ^ ((u64)a << 32) ^ (u64)b);
}
// one applied entry's contribution to the peer's apply digest (a pair of 32-bit sums, so order-free)
__host__ __device__ static inline u64 apply_word(u64 cmd, int idx) { return sm64(cmd ^ ((u64)(unsigned)idx << 40)); }
__host__ __device__ static inline u64 elem_hash(unsigned peer, unsigned field, unsigned idx, u64 v) {
  return sm64(sm64(((u64)peer << 48) ^ ((u64)field << 32) ^ idx) ^ v);
}

// Loads and stores share the vmcnt counter on gfx950: a wait for a rarely taken load placed after the branches
// merge would also drain every store in flight on the common path.  Rare loads therefore wait for their data
// right where they are issued, inside their branch, and the common path never waits for memory.
#define WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)   /* vmcnt(0) only */

// All lanes that exchange data through global memory in the tick kernel sit in ONE wavefront (a group never spans
// waves).  A wavefront's vector memory operations go through its CU's one vector L1 in issue order, so lanes of a
// wave see each other's stores without waiting for them: the fence only has to stop the compiler from moving
// accesses across it.
__device__ static inline void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// per-wave counter block in LDS
#define RG_CTR_SLOTS 56   // RG_N_COUNTERS public counters, then profiling slots (builds with -DRG_SECTION_COUNTS)
struct WaveCtr { unsigned c[RG_CTR_SLOTS]; };
#ifdef RG_SECTION_COUNTS
#define SECTION_COUNT(w, i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&(w)->c[RG_N_COUNTERS + (i)], 1u); } while (0)
#else
#define SECTION_COUNT(w, i) do { } while (0)
#endif
#ifdef RG_STAMPS   // diagnostic builds only: shader cycles per phase of a tick, summed per wave into slots 40..
#define STAMP(k) do { unsigned long long _n = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) atomicAdd(&w->c[40 + (k)], (unsigned)(_n - _stamp)); _stamp = _n; } while (0)
#define STAMP_INIT unsigned long long _stamp = __builtin_readcyclecounter()
#else
#define STAMP(k) do { } while (0)
#define STAMP_INIT do { } while (0)
#endif

enum { CD_SNAPTERM = 0, CD_SNAPVOTE, CD_SNAPLEN, CD_RNGCTR, CD_ROUNDSEQ, CD_AH_LO, CD_AH_HI, CD_N };

typedef __attribute__((address_space(3))) int lds_int;

// Registers of one peer.  CP = type of the pointer to the event-only words (LDS rows or a local array).
template <typename CP>
struct LaneT {
  int term, vote, commit, applied, logLen, timer, timeout, hbTimer, elNext, monNext;
  int snapSync;
  // caches of log words (rebuilt by k_derive after host writes): term of the last entry, first index from
  // which every entry carries the current term, term of the entry at commitIndex
  int lastTerm, firstOwn, cTerm;
  unsigned flags;
  // Words touched only by events (persist, draws, round numbering, apply digest) live outside the register
  // file: in the tick kernel in LDS, one 256-lane row per word (cs = 256), so any lane of the group can also read
  // a peer's word directly; in the one-thread-per-command kernels in a local array (cs = 1).
  CP cold; int cs;
  __device__ auto &cd(int f) const { return cold[f * cs]; }
  __device__ u64 applyHash() const { return ((u64)(unsigned)cd(CD_AH_HI) << 32) | (unsigned)cd(CD_AH_LO); }
  __device__ void setApplyHash(u64 v) const { cd(CD_AH_LO) = (int)(unsigned)v; cd(CD_AH_HI) = (int)(unsigned)(v >> 32); }

  __device__ int state() const { return (int)(flags & F_STATE_MASK); }
  __device__ void setState(int s) { flags = (flags & ~F_STATE_MASK) | (unsigned)s; }
  __device__ bool has(unsigned b) const { return (flags & b) != 0; }
  __device__ void set(unsigned b, bool on) { flags = on ? (flags | b) : (flags & ~b); }
  __device__ int cnt() const { return (int)((flags & F_CNT_MASK) >> F_CNT_SHIFT); }
  __device__ void setCnt(int c) { flags = (flags & ~F_CNT_MASK) | ((unsigned)c << F_CNT_SHIFT); }
  // GetState(), raft.go:90-100
  __device__ bool isLeader() const { return (flags & (F_DEAD | F_STOP | F_STATE_MASK)) == (unsigned)RG_LEADER; }

  __device__ void load(const DevPtrs &S, long long gl) {
    term = S.term[gl]; vote = S.vote[gl]; flags = (unsigned)S.flags[gl]; commit = S.commit[gl]; applied = S.applied[gl];
    logLen = S.logLen[gl]; timer = S.timer[gl]; timeout = S.timeout[gl]; hbTimer = S.hbTimer[gl]; elNext = S.elNext[gl];
    monNext = S.monNext[gl]; cd(CD_RNGCTR) = S.rngCtr[gl]; cd(CD_ROUNDSEQ) = S.roundSeq[gl];
    cd(CD_SNAPTERM) = S.snapTerm[gl]; cd(CD_SNAPVOTE) = S.snapVote[gl]; cd(CD_SNAPLEN) = S.snapLen[gl]; snapSync = S.snapSync[gl];
    lastTerm = S.lastTerm[gl]; firstOwn = S.firstOwn[gl]; cTerm = S.commitTerm[gl]; setApplyHash(S.applyHash[gl]);
  }
  __device__ void store(const DevPtrs &S, long long gl) const {
    S.term[gl] = term; S.vote[gl] = vote; S.flags[gl] = (int)flags; S.commit[gl] = commit; S.applied[gl] = applied;
    S.logLen[gl] = logLen; S.timer[gl] = timer; S.timeout[gl] = timeout; S.hbTimer[gl] = hbTimer; S.elNext[gl] = elNext;
    S.monNext[gl] = monNext; S.rngCtr[gl] = cd(CD_RNGCTR); S.roundSeq[gl] = cd(CD_ROUNDSEQ);
    S.snapTerm[gl] = cd(CD_SNAPTERM); S.snapVote[gl] = cd(CD_SNAPVOTE); S.snapLen[gl] = cd(CD_SNAPLEN); S.snapSync[gl] = snapSync;
    S.lastTerm[gl] = lastTerm; S.firstOwn[gl] = firstOwn; S.commitTerm[gl] = cTerm; S.applyHash[gl] = applyHash();
  }
};

// Addressing of one group's log rows.
// base is wave-uniform in the tick kernel (the block's first group), off the lane's
// 32-bit byte offset from it, so accesses use scalar-base + vector-offset addressing.
struct Rows {
  char *base; unsigned off; int rowBytes, cmdOff;
  __device__ int *termp(int i, int col) const { return (int *)(base + (size_t)(off + (unsigned)i * (unsigned)rowBytes + (unsigned)col * 4u)); }
  __device__ u64 *cmdp(int i, int col) const { return (u64 *)(base + (size_t)(off + (unsigned)i * (unsigned)rowBytes + (unsigned)cmdOff + (unsigned)col * 8u)); }
};
__device__ static inline Rows rows_of(char *arr, const DevCfg &c, int g) {
  Rows r; r.base = arr + (size_t)g * c.L * c.rowBytes; r.off = 0; r.rowBytes = c.rowBytes; r.cmdOff = c.cmdOff; return r;
}
// the arrays the tick loop itself touches; everything else is read once and written once
struct HotPtrs { char *logRows, *snapRows; int *nextIdx, *matchIdx; u64 *applyRec; };

__device__ static inline void ctr_add(WaveCtr *w, int which, unsigned v) { atomicAdd(&w->c[which], v); }

// persist(), raft.go:107-123: (term, vote, logs[0:commitIndex+1]).  Every statement of the reference that
// writes the log is followed by persist() before the peer's lock is released (raft.go:291-300, 354-358),
// so between handlers the persisted log equals the first snapLen rows of the log itself: the snapshot
// shares the log rows and persist() moves no entry.  snapSync only keeps the count of entries a copying
// implementation would have to write (RG_C_PERSIST_ENTRIES).  An imported blob (Persister swap) lives
// in the separate snapshot rows until the next persist().
template <class LN>
__device__ static inline void persist(LN &r, int L, WaveCtr *w) {
  int n = r.commit + 1;
  if (n > L) n = L;
  r.cd(CD_SNAPTERM) = r.term;
  r.cd(CD_SNAPVOTE) = r.vote;
  if (n > r.snapSync) {
    ctr_add(w, RG_C_PERSIST_ENTRIES, (unsigned)(n - r.snapSync));
    r.snapSync = n;
  }
  r.cd(CD_SNAPLEN) = n;
  r.flags = (r.flags | F_SNAPHAS) & ~F_SNAPEXT;
  ctr_add(w, RG_C_PERSISTS, 1);
}

// nextRandom(), raft.go:532-534, on the peer's seeded stream
template <class LN>
__device__ static inline int next_random(LN &r, u64 key, int peer) {
  const unsigned ctr = (unsigned)r.cd(CD_RNGCTR);
  r.cd(CD_RNGCTR) = (int)(ctr + 1u);
  u64 h = stream_hash(key, STREAM_ELECTION, (unsigned)peer, ctr);
  return (int)((unsigned)(h >> 32) % (unsigned)RG_TIMEOUT_SPAN_MS) + RG_TIMEOUT_BASE_MS;
}

// Make() + readPersist() + the head of election(), raft.go:700-731,129-152,538-547.
// The harness's start1 runs crash1 first (config.go:137), so the peer comes
// back disconnected.
template <class LN>
__device__ static inline void make_peer(LN &r, const Rows &lg, const Rows &sn, int col, int L, u64 key, int now, WaveCtr *w) {
  r.flags &= ~(F_DEAD | F_STATE_MASK | F_STOP | F_CNT_MASK | F_MONALIVE | F_FIRST | F_RESTART | F_CDIRTY);
  r.term = 0; r.vote = -1;
  *lg.termp(0, col) = 0; *lg.cmdp(0, col) = 0;
  r.logLen = 1; r.commit = 0; r.applied = 0; r.timer = now; r.hbTimer = now;
  r.lastTerm = 0; r.cTerm = 0;
  if (r.has(F_SNAPHAS) && r.cd(CD_SNAPLEN) >= 1) {
    r.term = r.cd(CD_SNAPTERM); r.vote = r.cd(CD_SNAPVOTE);
    int n = r.cd(CD_SNAPLEN);
    if (r.has(F_SNAPEXT))
      for (int i = r.snapSync; i < n; i++) { *lg.termp(i, col) = *sn.termp(i, col); *lg.cmdp(i, col) = *sn.cmdp(i, col); }
    if (n > r.snapSync) r.snapSync = n;
    r.logLen = n;
    persist(r, L, w);                     // raft.go:147 runs with commitIndex still 0
    r.commit = r.logLen - 1;              // raft.go:148
    wave_fence();
    r.lastTerm = *lg.termp(r.logLen - 1, col);
    WAIT_LOADS();
    r.cTerm = r.lastTerm;
  }
  r.monNext = 0;
  r.timeout = 0;
  r.firstOwn = r.logLen;
  if (!(r.term == 0 && col == 0)) r.timeout = next_random(r, key, col);
  r.elNext = now;
}

// Start(), raft.go:344-364.  Returns the index or -1.
template <class LN>
__device__ static inline int start_cmd(LN &r, const Rows &lg, const Rows &sn, int col, int L, u64 cmd, unsigned &gfl, WaveCtr *w) {
  if (!r.isLeader()) return -1;
  if (r.logLen >= L) { gfl |= RG_FLAG_LOG_OVERFLOW; return -1; }
  int idx = r.logLen;
  *lg.termp(idx, col) = r.term;
  *lg.cmdp(idx, col) = cmd;
  if (idx < r.snapSync) r.snapSync = idx;
  r.logLen = idx + 1;
  r.lastTerm = r.term;
  if (idx == r.commit) r.cTerm = r.term;       // only when commitIndex already ran past the log
  persist(r, L, w);
  ctr_add(w, RG_C_ENTRIES_STARTED, 1);
  return idx;
}

// ---- tentative handler results ------------------------------------------------
struct VoteDecision { int rTerm; bool touch, raise, grant; };
// RequestVote(), raft.go:200-232, evaluated without side effects.
template <class LN>
__device__ static inline VoteDecision decide_vote(const LN &r, int myLastTerm, int aTerm, int cand, int lli, int llt) {
  VoteDecision d;
  if (aTerm < r.term) { d.rTerm = r.term; d.touch = false; d.raise = false; d.grant = false; return d; }
  d.touch = true;
  d.raise = aTerm > r.term;
  int v = d.raise ? -1 : r.vote;
  d.rTerm = aTerm > r.term ? aTerm : r.term;
  d.grant = (v == -1 || v == cand) && (llt > myLastTerm || (llt == myLastTerm && lli >= r.commit));
  return d;
}
template <class LN>
__device__ static inline void commit_vote(LN &r, const VoteDecision &d, int aTerm, int cand, int now,
                                          const Rows &lg, const Rows &sn, int col, int L, WaveCtr *w) {
  if (!d.touch) return;
  if (d.raise) { r.term = aTerm; r.vote = -1; persist(r, L, w); }
  r.setState(RG_FOLLOWER);                                  // raft.go:216
  if (d.grant) { r.vote = cand; persist(r, L, w); r.timer = now; }
}

struct AppendDecision { int rTerm, ct, ci; bool touch, match; };
// AppendEntries(), raft.go:270-321, evaluated without side effects.
// myPrevTerm is logs[PrevLogIndex].Term of the receiver (any value if out of range).
template <class LN>
__device__ static inline AppendDecision decide_append(const LN &r, const Rows &lg, int col, int aTerm, int prev, int prevTerm, int myPrevTerm) {
  AppendDecision d; d.ct = 0; d.ci = 0; d.match = false;
  if (aTerm < r.term) { d.rTerm = r.term; d.touch = false; return d; }
  d.touch = true;
  d.rTerm = aTerm > r.term ? aTerm : r.term;
  if (prev + 1 <= r.logLen && myPrevTerm == prevTerm) { d.match = true; return d; }
  if (prev + 1 > r.logLen) { d.ct = -1; d.ci = r.logLen; return d; }
  d.ct = myPrevTerm;
  int idx = prev - 1;
  while (idx >= 0 && *lg.termp(idx, col) == d.ct) idx--;
  d.ci = idx + 1;
  return d;
}

// leader's back-off on a failed reply, raft.go:493-505; ownCol is the leader's column
__device__ static inline int leader_backoff(const Rows &lg, int ownCol, int nx, int ct, int ci) {
  if (ct >= 0) {
    int idx = nx - 1;
    while (*lg.termp(idx, ownCol) > ct) idx--;
    if (*lg.termp(idx, ownCol) == ct) return idx + 1;
    return ci;
  }
  return ci;
}

// ---------------------------------------------------------------- tick kernel
template <int PP>
struct Grp {
  int lane, base, peer;
  __device__ Grp() {
    lane = (int)(threadIdx.x & 63);
    base = lane & ~(PP - 1);
    peer = lane & (PP - 1);
  }
  __device__ int shfl(int v, int j) const { return __builtin_amdgcn_ds_bpermute((base | j) << 2, v); }
  __device__ unsigned shflu(unsigned v, int j) const { return (unsigned)__builtin_amdgcn_ds_bpermute((base | j) << 2, (int)v); }
  __device__ u64 shfl64(u64 v, int j) const {
    unsigned lo = shflu((unsigned)v, j), hi = shflu((unsigned)(v >> 32), j);
    return ((u64)hi << 32) | lo;
  }
  __device__ unsigned ballot(bool p) const { return (unsigned)(__builtin_amdgcn_ballot_w64(p) >> base) & ((1u << PP) - 1u); }
  __device__ int gmin(int v) const {
#pragma unroll
    for (int m = 1; m < PP; m <<= 1) { int o = __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, v); v = o < v ? o : v; }
    return v;
  }
  __device__ int gmax(int v) const {
#pragma unroll
    for (int m = 1; m < PP; m <<= 1) { int o = __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, v); v = o > v ? o : v; }
    return v;
  }
  __device__ int gsum(int v) const {
#pragma unroll
    for (int m = 1; m < PP; m <<= 1) v += __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, v);
    return v;
  }
  __device__ unsigned gor(unsigned v) const {
#pragma unroll
    for (int m = 1; m < PP; m <<= 1) v |= (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ m) << 2, (int)v);
    return v;
  }
};

// Dense variant for the tick kernel: a group's P peers sit in P adjacent lanes, 64/P whole groups per
// wave (12 for P = 5) and the 64 % P tail lanes idle, instead of padding every group to a power of two.
template <int P>
struct GrpN {
  static constexpr int GW = 64 / P;          // groups per wave
  int lane, base, peer;
  bool inWave;
  __device__ GrpN() {
    lane = (int)(threadIdx.x & 63);
    const int q = lane / P;
    base = q * P;
    peer = lane - base;
    inWave = q < GW;
  }
  __device__ int rot(int d) const { int x = peer + d; return base + (x >= P ? x - P : x); }
  __device__ int shfl(int v, int j) const { return __builtin_amdgcn_ds_bpermute((base + j) << 2, v); }
  __device__ unsigned shflu(unsigned v, int j) const { return (unsigned)__builtin_amdgcn_ds_bpermute((base + j) << 2, (int)v); }
  __device__ u64 shfl64(u64 v, int j) const {
    unsigned lo = shflu((unsigned)v, j), hi = shflu((unsigned)(v >> 32), j);
    return ((u64)hi << 32) | lo;
  }
  __device__ unsigned ballot(bool p) const { return (unsigned)(__builtin_amdgcn_ballot_w64(p) >> base) & ((1u << P) - 1u); }
  // idempotent reductions: doubling rotations cover the group in ceil(log2 P) steps
  __device__ int gmin(int v) const {
#pragma unroll
    for (int d = 1; d < P; d <<= 1) { int o = __builtin_amdgcn_ds_bpermute(rot(d) << 2, v); v = o < v ? o : v; }
    return v;
  }
  __device__ int gmax(int v) const {
#pragma unroll
    for (int d = 1; d < P; d <<= 1) { int o = __builtin_amdgcn_ds_bpermute(rot(d) << 2, v); v = o > v ? o : v; }
    return v;
  }
  __device__ unsigned gor(unsigned v) const {
#pragma unroll
    for (int d = 1; d < P; d <<= 1) v |= (unsigned)__builtin_amdgcn_ds_bpermute(rot(d) << 2, (int)v);
    return v;
  }
  __device__ int gsum(int v) const {
    if constexpr ((P & (P - 1)) == 0) {
#pragma unroll
      for (int d = 1; d < P; d <<= 1) v += __builtin_amdgcn_ds_bpermute(rot(d) << 2, v);
      return v;
    } else {
      int t = v;
#pragma unroll
      for (int d = 1; d < P; d++) t += __builtin_amdgcn_ds_bpermute(rot(d) << 2, v);
      return t;
    }
  }
};

__device__ static inline bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// The unreliable net's delays of one round, 8 bits per target (labrpc.go:222): one hash per round, so the
// sender can add up its waits without asking the targets.
__device__ static inline u64 round_delays(const DevCfg &c, u64 key, int s, unsigned seq) {
  if ((c.reqDrop | c.replyDrop | c.replyDelay) == 0) return 0;
  return stream_hash(key, STREAM_NET, 0xFFFF0000u | (unsigned)s, seq);
}
__device__ static inline int delay_of(u64 delays, int j) { return (int)((((unsigned)(delays >> (8 * j)) & 0xFFu) * (unsigned)RG_NET_DELAY_SPAN_MS) >> 8); }
// network outcome of one Call, labrpc.go:216-307: 0 lost, 1 handler ran / reply lost or late, 2 round trip;
// bit 2 (NET_WIRE): the net carried the reply, in time or not, and counted its bytes (labrpc.go:286-290)
#define NET_WIRE 4
__device__ static inline int net_outcome(const DevCfg &c, u64 key, bool reachable, int s, int j, unsigned seq, u64 delays) {
  if (!reachable) return 0;
  if ((c.reqDrop | c.replyDrop | c.replyDelay) == 0) return 2 | NET_WIRE;
  u64 h = stream_hash(key, STREAM_NET, ((unsigned)s << 8) | (unsigned)j, seq);
  if ((int)((unsigned)(h & 0xFFFFF) % 1000u) < c.reqDrop) return 0;
  if ((int)((unsigned)((h >> 20) & 0xFFFFF) % 1000u) < c.replyDrop) return 1;
  if (delay_of(delays, j) >= RG_RPC_WAIT_MS) return 1 | NET_WIRE;   // the handler runs, the sender has given up
  if ((int)((unsigned)((h >> 40) & 0xFFFFF) % 1000u) < c.replyDelay) return 1 | NET_WIRE;
  return 2 | NET_WIRE;
}
// How long a round holds its sender: 25 ms for every Call without a reply (raft.go:473,608), else the net's delay.
template <int P>
__device__ static inline int round_busy(unsigned calledM, unsigned okM, u64 delays) {
  int busy = 0;
#pragma unroll
  for (int j = 0; j < P; j++)
    if ((calledM >> j) & 1u) busy += ((okM >> j) & 1u) ? delay_of(delays, j) : RG_RPC_WAIT_MS;
  return busy;
}

// scheduler slack of the election loop's 50 ms sleep: 0..3 ms, seeded
__device__ static inline int sleep_slack(u64 key, int p, int now) {
  unsigned h = (unsigned)key ^ ((unsigned)p * 0x9E3779B1u) ^ ((unsigned)now * 0x85EBCA6Bu);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return (int)(h >> 30);
}
// first peer of a tick's actor order
__device__ static inline int turn_start(u64 key, int tick, int P) {
  unsigned h = (unsigned)tick * 0x9E3779B1u + (unsigned)key;
  h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
  return (int)(((h >> 16) * (unsigned)P) >> 16);
}

#ifndef RG_TICK_WAVES
#define RG_TICK_WAVES 4   // waves per SIMD the register allocation must allow
#endif
// BYTES: wire-byte accounting compiled in (rg_config.account_bytes); the plain instantiation carries none of it
template <int P, bool BYTES>
__global__ __launch_bounds__(256, RG_TICK_WAVES) void k_tick(const DevPtrs *Sp, HotPtrs H, DevCfg c, int t0, int nticks) {
  constexpr int PP = P <= 4 ? 4 : 8;        // stride of a group in the per-peer arrays and matrices in memory
  constexpr int NONE = 8;                   // "no such peer" for first-set-bit results
  constexpr int GW = GrpN<P>::GW;
  constexpr int GPB = 4 * GW;               // groups per 256-thread block
  __shared__ WaveCtr sctr[4];
  const GrpN<P> G;
  const int wave = (int)(threadIdx.x >> 6);
  WaveCtr *w = &sctr[wave];
  if (G.lane < RG_CTR_SLOTS) w->c[G.lane] = 0;
  __syncthreads();

  const int g0 = (int)blockIdx.x * GPB;                        // the block's first group: always < G
  const int g = g0 + wave * GW + (G.lane / P);
  const int peer = G.peer;
  const bool valid = G.inWave && g < c.G;
  const int L = c.L;
  const unsigned gLoc = valid ? (unsigned)(g - g0) : 0u;       // in-range group for address arithmetic of idle lanes
  const int gq = g0 + (int)gLoc;
  const long long gl = (long long)(G.inWave ? g : g0) * PP + (G.inWave ? peer : 0);   // slot in the per-peer arrays
  Rows lg, sn;
  lg.base = H.logRows + (size_t)g0 * L * c.rowBytes; lg.off = gLoc * (unsigned)L * (unsigned)c.rowBytes; lg.rowBytes = c.rowBytes; lg.cmdOff = c.cmdOff;
  sn = lg; sn.base = H.snapRows + (size_t)g0 * L * c.rowBytes;
  int *const nextBlk = H.nextIdx + (size_t)g0 * PP * PP;       // uniform; rows [group][owner][j]
  int *const matchBlk = H.matchIdx + (size_t)g0 * PP * PP;
  const unsigned mOff = gLoc * (unsigned)(PP * PP);
#define nextRow(i) nextBlk[mOff + (unsigned)(i)]
#define matchRow(i) matchBlk[mOff + (unsigned)(i)]
  u64 *const recBlk = H.applyRec + (size_t)g0 * L;
  const unsigned rOff = gLoc * (unsigned)L;
#define rec(i) recBlk[rOff + (unsigned)(i)]
  const u64 key = sm64(c.seedKey ^ (u64)(c.groupBase + g));

  __shared__ int coldLds[CD_N + (BYTES ? 4 : 2)][256];    // + two group words, + two per-lane byte-accounting sums
  // The last RING entries each peer wrote to its log, and the last RING records of each group, staged in LDS:
  // a follower copies a leader's fresh entries, and every peer applies its fresh entries, without a trip to
  // HBM.  Slot = index % RING, tagged with the index; HBM is written through and stays authoritative.
  constexpr int RING = 4;
  __shared__ int ringIdx[RING][256];
  __shared__ int ringTerm[RING][256];
  __shared__ u64 ringCmd[RING][256];
  __shared__ int recIdx[RING][GPB];
  __shared__ u64 recVal[RING][GPB];
  const int tid = (int)threadIdx.x;
#pragma unroll
  for (int q = 0; q < RING; q++) { ringIdx[q][tid] = -1; if (tid < GPB) recIdx[q][tid] = -1; }
  __syncthreads();
#define RING_PUSH(i_, tm_, cm_) do { const int _q = (i_) & (RING - 1); ringIdx[_q][tid] = (i_); ringTerm[_q][tid] = (tm_); ringCmd[_q][tid] = (cm_); } while (0)
  LaneT<lds_int *> r;
  r.cold = (lds_int *)&coldLds[0][threadIdx.x]; r.cs = 256;
  r.load(*Sp, gl);                       // tail lanes read a valid slot and never write it back
#define gStartSeq coldLds[CD_N][threadIdx.x]
#define gPartEpoch coldLds[CD_N + 1][threadIdx.x]
  gStartSeq = Sp->startSeq[gq]; gPartEpoch = Sp->partEpoch[gq];
  if constexpr (BYTES) { coldLds[CD_N + 2][threadIdx.x] = 0; coldLds[CD_N + 3][threadIdx.x] = 0; }
  int partNext = Sp->partNext[gq], recMax = Sp->recMax[gq];
  unsigned gfl = 0;
  // next ticks of the periodic control events (kept as counters: no division in the loop)
  // Within a launch each lane caches the nextIndex / matchIndex words ONE leader (cOwner) keeps for it; memory is
  // written through, so the matrices in HBM stay complete for other owners, dumps and digests.
  int cOwner = -1, cNext = 0, cMatch = 0;
  // stores to words another lane may read later are drained (vmcnt) only right before such a read
  bool storesPending = false;
#define SYNC_STORES() do { if (storesPending) { SECTION_COUNT(w, 7); wave_fence(); storesPending = false; } } while (0)
  bool chkLeaders = true;      // the one-leader check is re-evaluated only after something changed
  int nextStartTick = 0x7fffffff, nextChurnTick = 0x7fffffff, churnIt = 0;
  if (c.startPeriod > 0) {
    nextStartTick = c.startFirst;
    if (t0 > c.startFirst) nextStartTick += (t0 - c.startFirst + c.startPeriod - 1) / c.startPeriod * c.startPeriod;
  }
  if (c.churnPeriod > 0) {
    int from = t0 > 1 ? t0 : 1;
    churnIt = (from + c.churnPeriod - 1) / c.churnPeriod;
    nextChurnTick = churnIt * c.churnPeriod;
  }

  __builtin_amdgcn_s_waitcnt(0);         // nothing of the prologue is in flight when the loop starts
  for (int t = t0; t < t0 + nticks; t++) {
    const int now = t * RG_TICK_MS;
    STAMP_INIT;
    // ------------------------------------------------------------ 1. control
    if (c.partPeriod > 0 && wave_any(t == partNext)) chkLeaders = true;
    if (c.partPeriod > 0 && t == partNext) {
      unsigned ep = (unsigned)gPartEpoch;
      r.set(F_SIDE, (stream_hash(key, STREAM_PART, ep, (unsigned)peer + 1) & 1) != 0);
      u64 h = stream_hash(key, STREAM_PART, ep, 0);
      partNext = t + c.partPeriod + (c.partJitter > 0 ? (int)(h % (u64)c.partJitter) : 0);
      gPartEpoch++;
    }
    if (t == nextChurnTick) {                                        // test_test.go:893-919
      unsigned it = (unsigned)churnIt;
      churnIt++; nextChurnTick += c.churnPeriod;
      chkLeaders = true; storesPending = true;
      u64 h1 = stream_hash(key, STREAM_CHURN, it, 0), h2 = stream_hash(key, STREAM_CHURN, it, 1), h3 = stream_hash(key, STREAM_CHURN, it, 2);
      if ((int)((unsigned)(h1 & 0xFFFFF) % 1000u) < c.churnDisc && peer == (int)((h1 >> 32) % (u64)P)) r.flags &= ~F_CONNECTED;
      if ((int)((unsigned)(h2 & 0xFFFFF) % 1000u) < c.churnRestart && valid && peer == (int)((h2 >> 32) % (u64)P)) {
        if (r.has(F_DEAD)) {
          make_peer(r, lg, sn, peer, L, key, now, w);
#pragma unroll
          for (int q = 0; q < RING; q++) ringIdx[q][tid] = -1;     // the log may have been rebuilt from an imported blob
        }
        r.flags |= F_CONNECTED;
      }
      if ((int)((unsigned)(h3 & 0xFFFFF) % 1000u) < c.churnCrash && peer == (int)((h3 >> 32) % (u64)P) && !r.has(F_DEAD)) {
        r.flags &= ~F_CONNECTED;
        r.flags |= F_DEAD;
      }
    }
    if (t == nextStartTick) {
      nextStartTick += c.startPeriod;
      // client Starts: config.go:436-451 (first connected leader) / test_test.go:697-703 (every leader)
      bool cand = valid && r.isLeader() && (c.startMode == RG_START_ALL || r.has(F_CONNECTED));
      if (wave_any(cand)) {
        SECTION_COUNT(w, 5);
        bool room = cand && r.logLen < L;
        unsigned mAll = G.ballot(cand), mRoom = G.ballot(room);
        unsigned below = (1u << peer) - 1u;
        if (c.startMode == RG_START_ALL) {
          int seq = gStartSeq + __popc(mRoom & below);
          bool tried = cand && seq < c.startMax;
          if (tried) {
            u64 cmd = stream_hash(key, STREAM_CMD, 0, (unsigned)seq);
            const int at = start_cmd(r, lg, sn, peer, L, cmd, gfl, w);
            if (at >= 0) RING_PUSH(at, r.term, cmd);
          }
          int acc = __popc(mRoom);
          int left = c.startMax - gStartSeq;
          gStartSeq += acc < left ? acc : (left > 0 ? left : 0);
        } else if (gStartSeq < c.startMax) {
          // peers without room are tried (and flagged) until one accepts
          int wIdx = mRoom ? (__ffs((int)mRoom) - 1) : NONE;
          if (cand && peer <= wIdx) {
            u64 cmd = stream_hash(key, STREAM_CMD, 0, (unsigned)gStartSeq);
            const int at = start_cmd(r, lg, sn, peer, L, cmd, gfl, w);
            if (at >= 0) RING_PUSH(at, r.term, cmd);
          }
          if (mRoom) gStartSeq++;
          (void)mAll;
        }
        storesPending = true;
      }
    }

    STAMP(0);
    // ------------------------------------------------------------ 2. actors
    {
      const bool live = valid && !r.has(F_DEAD);
      const bool elDue = live && r.elNext <= now;
      const bool monDue = live && r.has(F_MONALIVE) && r.monNext <= now;
      int start = 0, myPos = 0;
      bool pend = false, specEl = false, specMon = false, pendLocal = false;
      if (elDue) {
        int st = r.state();
        if (st == RG_FOLLOWER) {
          if (now - r.timer > r.timeout) { pend = true; pendLocal = true; }
          else r.elNext = now + RG_ELECTION_PERIOD_MS + sleep_slack(key, peer, now);   // nothing later in this tick can make it fire
        } else if (st == RG_CANDIDATE) pend = true;
        else specEl = true;
      }
      if (monDue) {
        if (!r.isLeader()) {
          if (r.state() == RG_CANDIDATE) pend = true;      // its own round may win first
          else { r.flags &= ~F_MONALIVE; r.monNext = 0; }  // raft.go:526-528
        } else {
          int since = now - r.hbTimer;
          bool gate = r.has(F_FIRST) || since >= RG_HB_IDLE_MS || (r.commit < r.logLen - 1 && since >= RG_HB_BACKLOG_MS);
          if (gate || r.has(F_CDIRTY)) pend = true; else specMon = true;
        }
      }
      if (pend) { specEl = false; specMon = false; }
      // A timed-out follower only turns candidate (raft.go:557-562): no RPC.  When no peer of its group can run a
      // round in this tick nothing can reach it before its turn, so it need not wait for it.
      if (wave_any(pendLocal)) {
        const bool grpRound = G.ballot(pend && !pendLocal) != 0;
        if (pendLocal && !grpRound) {
          r.setState(RG_CANDIDATE);
          r.timeout = next_random(r, key, peer);
          ctr_add(w, RG_C_TIMEOUTS, 1);
          r.elNext = now + RG_ELECTION_PERIOD_MS + sleep_slack(key, peer, now);
          pend = false;
        }
      }
      STAMP(1);
      if (wave_any(pend)) { start = turn_start(key, t, P); myPos = peer - start + (peer < start ? P : 0); }

      while (wave_any(pend)) {
        SECTION_COUNT(w, 0);
        chkLeaders = true;
        const unsigned pm = G.ballot(pend);
        const unsigned pmHi = pm >> start;                   // cyclic order from this tick's start
        const int s = pmHi ? (start + __ffs((int)pmHi) - 1) : (pm ? (__ffs((int)pm) - 1) : -1);   // this group's acting peer
        const int sPos = s - start + (s < start ? P : 0);
        const bool isS = pend && peer == s;
        const int sIdx = s < 0 ? 0 : s;
        // ---------------- election loop iteration, raft.go:548-668
        bool wantRV = false;
        if (isS && r.elNext <= now) {
          int st = r.state();
          if (st == RG_FOLLOWER) {
            if (now - r.timer > r.timeout) {
              r.setState(RG_CANDIDATE);
              r.timeout = next_random(r, key, peer);
              ctr_add(w, RG_C_TIMEOUTS, 1);
            }
            r.elNext = now + RG_ELECTION_PERIOD_MS + sleep_slack(key, peer, now);
          } else if (st == RG_CANDIDATE) {
            wantRV = true;
          } else {
            r.elNext = now + RG_ELECTION_PERIOD_MS + RG_ELECTION_LEADER_EXTRA_MS + sleep_slack(key, peer, now);
          }
        }
        STAMP(5);
        if (wave_any(wantRV)) {
          SECTION_COUNT(w, 1);
          // candidate prologue, raft.go:566-590
          int aLli = 0, aLlt = 0;
          if (wantRV) {
            r.timeout = RG_CANDIDATE_WINDOW_MS;
            r.timer = now;
            if (!r.has(F_RESTART)) r.term++; else r.flags &= ~F_RESTART;
            r.vote = peer;
            persist(r, L, w);
            aLli = r.commit;
            aLlt = r.cTerm;                                  // logs[commitIndex].Term, raft.go:583
            persist(r, L, w);
            ctr_add(w, RG_C_RV_ROUNDS, 1);
          }
          const bool inR = G.ballot(wantRV) != 0;
          const int T = G.shfl(r.term, sIdx), lli = G.shfl(aLli, sIdx), llt = G.shfl(aLlt, sIdx);
          const unsigned seq = (unsigned)r.cold[CD_ROUNDSEQ * 256 + (sIdx - peer)];   // the sender's word, read in place
          const unsigned fS = G.shflu(r.flags, sIdx);
          if (wantRV) r.cd(CD_ROUNDSEQ) = (int)(seq + 1u);
          const bool tgt = inR && valid && peer != s;
          const bool reach = tgt && (fS & F_CONNECTED) && r.has(F_CONNECTED) && ((fS ^ r.flags) & F_SIDE) == 0 && !r.has(F_DEAD);
          const u64 delays = round_delays(c, key, sIdx, seq);
          const int outW = tgt ? net_outcome(c, key, reach, s, peer, seq, delays) : 0;
          const int out = outW & 3;
          VoteDecision d; d.rTerm = 0; d.touch = d.raise = d.grant = false;
          if (out != 0) d = decide_vote(r, r.cTerm, T, s, lli, llt);
          const unsigned sdm = G.ballot(out == 2 && d.rTerm > T);      // raft.go:622
          const int b = sdm ? (__ffs((int)sdm) - 1) : NONE;
          const bool called = tgt && peer <= b, ran = out != 0 && peer <= b, ok = out == 2 && peer <= b;
          if (ran) {
            const bool wasLead = r.state() == RG_LEADER;
            commit_vote(r, d, T, s, now, lg, sn, peer, L, w);
            if (wasLead && d.touch && myPos > sPos && (specEl || specMon)) { pend = true; specEl = false; specMon = false; }
          }
          const unsigned calledM = G.ballot(called), ranM = G.ballot(ran), okM = G.ballot(ok), grantM = G.ballot(ok && d.grant);
          if constexpr (BYTES) atomicAdd((unsigned *)&coldLds[CD_N + 2][tid], called ? (unsigned)RG_VOTE_ARGS_BYTES + ((ran && (outW & NET_WIRE)) ? (unsigned)RG_VOTE_REPLY_BYTES : 0u) : 0u);
          const int busy = round_busy<P>(calledM, okM, delays);
          bool won = false;
          if (wantRV) {
            ctr_add(w, RG_C_RPC_SENT, (unsigned)__popc(calledM));
            ctr_add(w, RG_C_RPC_DELIVERED, (unsigned)__popc(ranM));
            ctr_add(w, RG_C_RPC_REPLIED, (unsigned)__popc(okM));
            int voteCnt = 1 + __popc(grantM);
            if (b < NONE) {                                          // raft.go:622-627: term is not adopted
              r.setState(RG_FOLLOWER);
              r.vote = -1;
              persist(r, L, w);
              ctr_add(w, RG_C_STEPDOWNS, 1);
              r.timeout = next_random(r, key, peer);                // raft.go:639
            } else if (2 * voteCnt > P) {                          // raft.go:641-658
              won = true;
              r.flags &= ~(F_STOP | F_CNT_MASK);
              r.setState(RG_LEADER);
              r.hbTimer = now + busy;
              r.flags |= F_MONALIVE | F_FIRST | F_CDIRTY;
              r.monNext = now + busy;
              int f = r.logLen;
              if (r.lastTerm == r.term) { while (f > 0 && *lg.termp(f - 1, peer) == r.term) f--; }
              r.firstOwn = f;
              ctr_add(w, RG_C_ELECTIONS_WON, 1);
            } else {
              r.flags |= F_RESTART;
            }
            r.elNext = now + busy + RG_ELECTION_PERIOD_MS + sleep_slack(key, peer, now);
          }
          const bool grpWon = G.ballot(won) != 0;
          int lenW = 0;
          if (wave_any(won)) lenW = G.shfl(r.logLen, sIdx);
          if (grpWon && valid) { nextRow(sIdx * PP + peer) = lenW; matchRow(sIdx * PP + peer) = 0; cOwner = s; cNext = lenW; cMatch = 0; }
          if (wave_any(grpWon)) storesPending = true;
        }
        STAMP(6);
        // ---------------- leader loop, raft.go:395-529, until it sleeps or exits
        bool monAct = isS && r.has(F_MONALIVE) && r.monNext <= now;
        for (;;) {
          bool wantAE = false;
          // commit rule, raft.go:400-416, in closed form: the largest N held by a majority is the floor(P/2)-th
          // largest follower matchIndex.  Each follower ranks its own value against the others' (from its cache
          // of this leader's row, filled from memory on a miss) and the group maximum of the qualifying values
          // is N.  The loop of the reference reaches N only if every entry from N up carries the current term:
          // N >= firstOwn.
          int Nstar = 0;
          const bool needCR = monAct && r.isLeader() && r.has(F_CDIRTY);
          if (wave_any(needCR)) {
            const bool fol = G.ballot(needCR) != 0 && valid && peer != s;
            if (wave_any(fol && cOwner != s)) {
              SYNC_STORES();
              if (fol && cOwner != s) { cNext = nextRow(sIdx * PP + peer); cMatch = matchRow(sIdx * PP + peer); cOwner = s; WAIT_LOADS(); }
            }
            const int mv = fol ? cMatch : -1;
            int ge = 0;
#pragma unroll
            for (int u = 0; u < P; u++) ge += G.shfl(mv, u) >= mv ? 1 : 0;
            Nstar = G.gmax((fol && ge >= P / 2) ? mv : 0);
          }
          if (monAct) {
            if (!r.isLeader()) { r.flags &= ~F_MONALIVE; r.monNext = 0; monAct = false; }
            else {
              if (r.has(F_CDIRTY)) {
                r.flags &= ~F_CDIRTY;
                int N = Nstar;
                if (N > r.logLen - 1) N = r.logLen - 1;
                if (N > r.commit && N >= r.firstOwn) {
                  ctr_add(w, RG_C_ENTRIES_COMMITTED, (unsigned)(N - r.commit));
                  r.commit = N;
                  r.cTerm = r.term;                                  // N >= firstOwn
                  r.flags |= F_FIRST;
                  persist(r, L, w);
                }
              }
              int since = now - r.hbTimer;
              if (r.has(F_FIRST) || since >= RG_HB_IDLE_MS || (r.commit < r.logLen - 1 && since >= RG_HB_BACKLOG_MS)) {
                r.flags &= ~F_FIRST;
                wantAE = true;
                ctr_add(w, RG_C_AE_ROUNDS, 1);
              } else {
                r.monNext = now + RG_MONITOR_POLL_MS;
                monAct = false;
              }
            }
          }
          SECTION_COUNT(w, 2);
          STAMP(7);
          if (!wave_any(wantAE)) break;
          SECTION_COUNT(w, 3);
          // ---------------- one replication round, raft.go:423-512
          const bool inR = G.ballot(wantAE) != 0;
          const int T = G.shfl(r.term, sIdx), lenS = G.shfl(r.logLen, sIdx), commitS = G.shfl(r.commit, sIdx);
          const int lastTermS = G.shfl(r.lastTerm, sIdx), firstOwnS = G.shfl(r.firstOwn, sIdx), cTermS = G.shfl(r.cTerm, sIdx);
          const unsigned seq = (unsigned)r.cold[CD_ROUNDSEQ * 256 + (sIdx - peer)];
          const unsigned fS = G.shflu(r.flags, sIdx);
          if (wantAE) r.cd(CD_ROUNDSEQ) = (int)(seq + 1u);
          const bool tgt = inR && valid && peer != s;
          if (wave_any(tgt && cOwner != s)) {
            SECTION_COUNT(w, 8);
            SYNC_STORES();
            if (tgt && cOwner != s) { cNext = nextRow(sIdx * PP + peer); cMatch = matchRow(sIdx * PP + peer); cOwner = s; WAIT_LOADS(); }
          }
          const int nx = tgt ? cNext : 0;
          const unsigned ovm = G.ballot(tgt && nx > lenS);           // raft.go:451-454 ends the round
          const int firstOv = ovm ? (__ffs((int)ovm) - 1) : NONE;
          const bool tgt2 = tgt && peer < firstOv;
          const bool reach = tgt2 && (fS & F_CONNECTED) && r.has(F_CONNECTED) && ((fS ^ r.flags) & F_SIDE) == 0 && !r.has(F_DEAD);
          const u64 delays = round_delays(c, key, sIdx, seq);
          const int outW = tgt2 ? net_outcome(c, key, reach, s, peer, seq, delays) : 0;
          const int out = outW & 3;
          const int prev = nx - 1, k = lenS - nx;
          AppendDecision d; d.rTerm = 0; d.ct = 0; d.ci = 0; d.touch = d.match = false;
          int prevTerm = 0;
          const int sTid = tid - peer + sIdx;                        // the sender's lane in the block
          // logs[Prev].Term on both sides: from the lanes' caches, else from the rings, else from HBM
          int myPrevTerm = -2;
          bool needS = false, needM = false;
          if (out != 0) {
            prevTerm = prev == lenS - 1 ? lastTermS : T;
            needS = prev != lenS - 1 && prev < firstOwnS;
            if (prev == r.logLen - 1) myPrevTerm = r.lastTerm; else needM = prev < r.logLen - 1;
            const int pq = prev & (RING - 1);
            if (needS && ringIdx[pq][sTid] == prev) { prevTerm = ringTerm[pq][sTid]; needS = false; }
            if (needM && ringIdx[pq][tid] == prev) { myPrevTerm = ringTerm[pq][tid]; needM = false; }
          }
          if (wave_any(needS || needM)) {
            SECTION_COUNT(w, 9);
            SYNC_STORES();
            if (needS || needM) {
              if (needS) prevTerm = *lg.termp(prev, sIdx);
              if (needM) myPrevTerm = *lg.termp(prev, peer);
              WAIT_LOADS();
            }
          }
          if (out != 0) d = decide_append(r, lg, peer, T, prev, prevTerm, myPrevTerm);
          const unsigned sdm = G.ballot(out == 2 && d.rTerm > T);    // raft.go:482
          const int b = sdm ? (__ffs((int)sdm) - 1) : NONE;
          const bool called = tgt2 && peer <= b, ran = out != 0 && peer <= b, ok = out == 2 && peer <= b;
          // Entries the sender wrote recently are copied from its ring.  A longer or older run (a follower catching
          // up) is copied by the whole wave, one row per lane, so that its cost is one trip to HBM whatever its length.
          bool viaHbm = false;
          if (ran && d.touch && d.match && k > 0) {
            viaHbm = k > RING;
#pragma unroll
            for (int q = 0; q < RING; q++) {               // slot q would hold the one index of nx..nx+3 that is q mod RING
              const int want = nx + ((q - nx) & (RING - 1));
              if (want < lenS && ringIdx[q][sTid] != want) viaHbm = true;
            }
          }
          {
            u64 hm = __builtin_amdgcn_ballot_w64(viaHbm);
            const bool scan = ok && b != peer && d.touch && !d.match;
            if (hm) SECTION_COUNT(w, 10);
            if (wave_any(scan)) SECTION_COUNT(w, 11);
            if (hm || wave_any(scan)) SYNC_STORES();
            while (hm) {
              const int src = (int)__builtin_ctzll(hm);
              hm &= hm - 1;
              const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src);
              const int nxB = __builtin_amdgcn_readlane(nx, src), lenB = __builtin_amdgcn_readlane(lenS, src);
              const unsigned sCol = (unsigned)__builtin_amdgcn_readlane(sIdx, src), tCol = (unsigned)__builtin_amdgcn_readlane(peer, src);
              const int tTid = (tid & ~63) + src;
              for (int i = nxB + G.lane; i < lenB; i += 64) {
                char *row = lg.base + (size_t)(offB + (unsigned)i * (unsigned)lg.rowBytes);
                const int tm = *(int *)(row + sCol * 4u);
                const u64 cm = *(u64 *)(row + (unsigned)lg.cmdOff + sCol * 8u);
                *(int *)(row + tCol * 4u) = tm;
                *(u64 *)(row + (unsigned)lg.cmdOff + tCol * 8u) = cm;
                if (i >= lenB - RING) { const int q = i & (RING - 1); ringIdx[q][tTid] = i; ringTerm[q][tTid] = tm; ringCmd[q][tTid] = cm; }
              }
            }
          }
          if (ran && d.touch) {
            const bool wasLead = r.state() == RG_LEADER;
            r.setState(RG_FOLLOWER);
            r.vote = s;                                              // raft.go:281
            if (r.term < T) { r.term = T; persist(r, L, w); }
            r.timer = now;
            if (d.match) {
              if (!viaHbm) {
                for (int i = nx; i < lenS; i++) {                    // at most RING entries, all in the sender's ring
                  const int q = i & (RING - 1);
                  const int tm = ringTerm[q][sTid]; const u64 cm = ringCmd[q][sTid];
                  *lg.termp(i, peer) = tm; *lg.cmdp(i, peer) = cm;
                  RING_PUSH(i, tm, cm);
                }
              }
              if (k > 0 && nx < r.snapSync) r.snapSync = nx;
              r.logLen = lenS;
              r.lastTerm = k > 0 ? lastTermS : prevTerm;
              const int oldCommit = r.commit;
              if (commitS > r.commit) r.commit = (r.logLen - 1 < commitS) ? r.logLen - 1 : commitS;
              if (r.commit > r.logLen - 1) gfl |= RG_FLAG_REF_PANIC;
              if (r.commit != oldCommit || nx <= r.commit) {          // the entry at commitIndex is new or was rewritten
                if (r.commit == lenS - 1) r.cTerm = r.lastTerm;
                else if (r.commit == commitS && commitS < lenS) r.cTerm = cTermS;
                else { wave_fence(); r.cTerm = *lg.termp(r.commit < L ? r.commit : L - 1, peer); WAIT_LOADS(); }
              }
              persist(r, L, w);
              ctr_add(w, RG_C_ENTRIES_REPLICATED, (unsigned)k);
            } else {
              ctr_add(w, RG_C_AE_CONFLICTS, 1);
            }
            if (wasLead && myPos > sPos && (specEl || specMon)) { pend = true; specEl = false; specMon = false; }
          }
          // the leader's handling of each reply, done by the replying lane
          if (ok && b != peer) {
            if (d.match) { cNext = lenS; cMatch = lenS - 1; matchRow(sIdx * PP + peer) = cMatch; }
            else cNext = leader_backoff(lg, sIdx, nx, d.ct, d.ci);   // a stale-term reply would have ended the round
            nextRow(sIdx * PP + peer) = cNext;
          }
          const unsigned calledM = G.ballot(called), ranM = G.ballot(ran), okM = G.ballot(ok);
          if constexpr (BYTES) {   // wire bytes: summed per target lane in LDS words of its own, added to the counters when the launch ends
            atomicAdd((unsigned *)&coldLds[CD_N + 2][tid], called ? (unsigned)RG_APPEND_ARGS_BYTES + ((ran && (outW & NET_WIRE)) ? (unsigned)RG_APPEND_REPLY_BYTES : 0u) : 0u);
            atomicAdd((unsigned *)&coldLds[CD_N + 3][tid], (called && k > 0) ? (unsigned)k : 0u);
          }
          const int busy = round_busy<P>(calledM, okM, delays);
          int rTermB = 0;
          if (wave_any(b < NONE)) rTermB = G.shfl(d.rTerm, b < NONE ? b : 0);
          if (wantAE) {
            ctr_add(w, RG_C_RPC_SENT, (unsigned)__popc(calledM));
            ctr_add(w, RG_C_RPC_DELIVERED, (unsigned)__popc(ranM));
            ctr_add(w, RG_C_RPC_REPLIED, (unsigned)__popc(okM));
            ctr_add(w, RG_C_AE_DELIVERED, (unsigned)__popc(ranM));
            if (b < NONE) {                                            // raft.go:482-487: voteFor is kept
              r.setState(RG_FOLLOWER);
              r.term = rTermB;
              persist(r, L, w);
              ctr_add(w, RG_C_STEPDOWNS, 1);
            }
            int active = 1 + __popc(okM);
            if (active == 1 && r.cnt() <= 3) {                       // raft.go:514-520
              int cn = r.cnt() + 1;
              r.setCnt(cn);
              if (cn > 3) r.flags |= F_STOP;
            }
            r.hbTimer = now + busy;                                  // raft.go:521
            r.flags |= F_CDIRTY;
            if (busy > 0) { r.monNext = now + busy; monAct = false; }   // the loop goes on once the waits are over
          }
          storesPending = true;
          STAMP(8);
        }
        if (isS) pend = false;
      }
      if (specEl) r.elNext = now + RG_ELECTION_PERIOD_MS + RG_ELECTION_LEADER_EXTRA_MS + sleep_slack(key, peer, now);
      if (specMon) r.monNext = now + RG_MONITOR_POLL_MS;
    }

    STAMP(2);
    // ------------------------------------------------------------ 3. apply, raft.go:676-686 + config.go:170-194
    // The reference's tester sees peers 0..P-1 deliver in turn: peer p checks its entries up to the newest index
    // recorded before its turn (pm = the larger of the record's end and the commitIndex of every lower peer)
    // against the record, and records the rest.  Ranges of different peers that are recorded never overlap and the
    // digest is a sum, so all peers record first and then all peers check, each range in one go.
    {
      const bool act = valid && !r.has(F_DEAD) && r.commit > r.applied;
      if (wave_any(act)) {
        SECTION_COUNT(w, 4);
        const int cA = act ? r.commit : 0;
        int pm = recMax, newMax = recMax;
        if (wave_any(cA > recMax)) {
#pragma unroll
          for (int u = 0; u < P; u++) {
            const int v = G.shfl(cA, u);
            if (u < peer) pm = v > pm ? v : pm;
            newMax = v > newMax ? v : newMax;
          }
        }
        const int cmpHi = cA < pm ? cA : pm;
        const int nCmp = (act && cmpHi > r.applied) ? cmpHi - r.applied : 0;       // checks (applied, cmpHi]
        const int recLo = r.applied > pm ? r.applied : pm;
        const int nRec = (act && r.commit > recLo) ? r.commit - recLo : 0;         // records (recLo, commit]
        if (nRec > 0 && r.applied > pm) gfl |= RG_FLAG_APPLY_ORDER;                // config.go:184-187
        if (act && r.commit >= r.logLen) gfl |= RG_FLAG_REF_PANIC;                 // raft.go:679 would index past the log
        unsigned accLo = 0, accHi = 0;
        bool bad = false;
        // ---- record
        if (wave_any(nRec > 0)) {
          bool slow = nRec > RING;
#pragma unroll
          for (int q = 0; q < RING; q++) {
            const int want = recLo + 1 + ((q - recLo - 1) & (RING - 1));
            if (want <= r.commit && nRec > 0 && ringIdx[q][tid] != want) slow = true;
          }
          u64 sm = __builtin_amdgcn_ballot_w64(slow);
          if (nRec > 0 && !slow) {
            for (int j = 0; j < nRec; j++) {
              const int i = recLo + 1 + j, q = i & (RING - 1);
              const u64 cmd = ringCmd[q][tid];
              rec(i < L ? i : L - 1) = cmd;
              if (i > newMax - RING) { recIdx[q][gLoc] = i; recVal[q][gLoc] = cmd; }
              const u64 wd = apply_word(cmd, i);
              accLo += (unsigned)wd; accHi += (unsigned)(wd >> 32);
            }
          }
          if (sm) {                                   // long or old runs: the whole wave, one entry per lane
            SECTION_COUNT(w, 13);
            SYNC_STORES();
            do {
              const int src = (int)__builtin_ctzll(sm);
              sm &= sm - 1;
              const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src), rOffB = (unsigned)__builtin_amdgcn_readlane((int)rOff, src);
              const int lo = __builtin_amdgcn_readlane(recLo, src), n = __builtin_amdgcn_readlane(nRec, src);
              const int top = __builtin_amdgcn_readlane(newMax, src), gSrc = __builtin_amdgcn_readlane((int)gLoc, src);
              const unsigned col = (unsigned)__builtin_amdgcn_readlane(peer, src);
              const int tTid = (tid & ~63) + src;
              for (int j = G.lane; j < n; j += 64) {
                const int i = lo + 1 + j, ic = i < L ? i : L - 1, q = i & (RING - 1);
                const u64 cmd = *(u64 *)(lg.base + (size_t)(offB + (unsigned)ic * (unsigned)lg.rowBytes + (unsigned)lg.cmdOff + col * 8u));
                recBlk[rOffB + (unsigned)ic] = cmd;
                if (i > top - RING) { recIdx[q][gSrc] = i; recVal[q][gSrc] = cmd; }
                const u64 wd = apply_word(cmd, i);
                atomicAdd((unsigned *)&coldLds[CD_AH_LO][tTid], (unsigned)wd);
                atomicAdd((unsigned *)&coldLds[CD_AH_HI][tTid], (unsigned)(wd >> 32));
              }
            } while (sm);
          }
          recMax = newMax;
          storesPending = true;
        }
        // ---- check
        if (wave_any(nCmp > 0)) {
          bool slow = nCmp > RING;
#pragma unroll
          for (int q = 0; q < RING; q++) {
            const int want = r.applied + 1 + ((q - r.applied - 1) & (RING - 1));
            if (want <= cmpHi && nCmp > 0 && !(ringIdx[q][tid] == want && recIdx[q][gLoc] == want)) slow = true;
          }
          u64 sm = __builtin_amdgcn_ballot_w64(slow);
          if (nCmp > 0 && !slow) {
            for (int j = 0; j < nCmp; j++) {
              const int i = r.applied + 1 + j, q = i & (RING - 1);
              const u64 cmd = ringCmd[q][tid];
              bad |= cmd != recVal[q][gLoc];
              const u64 wd = apply_word(cmd, i);
              accLo += (unsigned)wd; accHi += (unsigned)(wd >> 32);
            }
          }
          if (sm) {
            SECTION_COUNT(w, 12);
            SYNC_STORES();
            do {
              const int src = (int)__builtin_ctzll(sm);
              sm &= sm - 1;
              const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src), rOffB = (unsigned)__builtin_amdgcn_readlane((int)rOff, src);
              const int a = __builtin_amdgcn_readlane(r.applied, src), n = __builtin_amdgcn_readlane(nCmp, src);
              const unsigned col = (unsigned)__builtin_amdgcn_readlane(peer, src);
              const int tTid = (tid & ~63) + src;
              bool differs = false;
              for (int j = G.lane; j < n; j += 64) {
                const int i = a + 1 + j, ic = i < L ? i : L - 1;
                const u64 cmd = *(u64 *)(lg.base + (size_t)(offB + (unsigned)ic * (unsigned)lg.rowBytes + (unsigned)lg.cmdOff + col * 8u));
                const u64 seen = recBlk[rOffB + (unsigned)ic];
                differs |= cmd != seen;
                const u64 wd = apply_word(cmd, i);
                atomicAdd((unsigned *)&coldLds[CD_AH_LO][tTid], (unsigned)wd);
                atomicAdd((unsigned *)&coldLds[CD_AH_HI][tTid], (unsigned)(wd >> 32));
              }
              if (wave_any(differs) && G.lane == src) bad = true;
            } while (sm);
          }
        }
        if (act) {
          if (bad) gfl |= RG_FLAG_STATE_MACHINE;
          ctr_add(w, RG_C_ENTRIES_APPLIED, (unsigned)(r.commit - r.applied));
          r.applied = r.commit;
          atomicAdd((unsigned *)&coldLds[CD_AH_LO][tid], accLo);
          atomicAdd((unsigned *)&coldLds[CD_AH_HI][tid], accHi);
        }
      }
    }

    STAMP(3);
    // ------------------------------------------------------------ 4. one leader per term, config.go:307-320
    {
      const bool lead = chkLeaders && valid && r.has(F_CONNECTED) && r.isLeader();
      const unsigned lm = G.ballot(lead);
      chkLeaders = false;
      if (wave_any(lead && __popc(lm) > 1)) {
        SECTION_COUNT(w, 6);
#pragma unroll
        for (int u = 0; u < P; u++) {
          int tu = G.shfl(r.term, u);
          if (lead && u != peer && ((lm >> u) & 1u) && tu == r.term) gfl |= RG_FLAG_ELECTION_SAFETY;
        }
      }
    }
    STAMP(4);
  }

#undef SYNC_STORES
#undef RING_PUSH
  const int startSeqOut = gStartSeq, partEpochOut = gPartEpoch;
  if constexpr (BYTES) {
    if (coldLds[CD_N + 2][threadIdx.x]) ctr_add(w, RG_C_RPC_FIXED_BYTES, (unsigned)coldLds[CD_N + 2][threadIdx.x]);
    if (coldLds[CD_N + 3][threadIdx.x]) ctr_add(w, RG_C_AE_ENTRIES_SENT, (unsigned)coldLds[CD_N + 3][threadIdx.x]);
  }
#undef gStartSeq
#undef gPartEpoch
#undef nextRow
#undef matchRow
#undef rec
  if (G.inWave) r.store(*Sp, gl);
  gfl = G.gor(gfl);
  if (valid && peer == 0) {
    Sp->startSeq[g] = startSeqOut; Sp->partNext[g] = partNext; Sp->partEpoch[g] = partEpochOut; Sp->recMax[g] = recMax;
    if (gfl) Sp->gflags[g] |= gfl;
  }
  __syncthreads();
  if (G.lane < RG_CTR_SLOTS && w->c[G.lane]) atomicAdd((unsigned long long *)&Sp->counters[G.lane], (unsigned long long)w->c[G.lane]);
}

// ---------------------------------------------------------------- host-verb kernels
// One thread per command; commands of one launch touch distinct groups.
enum { OP_START = 0, OP_CONNECT, OP_DISCONNECT, OP_SIDE0, OP_SIDE1, OP_CRASH, OP_RESTART, OP_GETSTATE };
struct HostCmd { int op, group, peer, pad; u64 arg; int r0, r1, r2, r3; };

__global__ void k_host_cmds(DevPtrs S, DevCfg c, int PP, HostCmd *cmds, int n, int now) {
  __shared__ WaveCtr sctr;
  if (threadIdx.x < RG_N_COUNTERS) sctr.c[threadIdx.x] = 0;
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    HostCmd cm = cmds[i];
    const int g = cm.group, p = cm.peer;
    const long long gl = (long long)g * PP + p;
    const Rows lg = rows_of(S.logRows, c, g), sn = rows_of(S.snapRows, c, g);
    const u64 key = sm64(c.seedKey ^ (u64)(c.groupBase + g));
    int coldLocal[CD_N];
    LaneT<int *> r; r.cold = coldLocal; r.cs = 1; r.load(S, gl);
    unsigned gfl = 0;
    switch (cm.op) {
    case OP_START: {
      cm.r1 = r.term; cm.r2 = r.isLeader() ? 1 : 0;
      cm.r0 = start_cmd(r, lg, sn, p, c.L, cm.arg, gfl, &sctr);
      if (cm.r0 < 0) cm.r2 = 0;
      break;
    }
    case OP_CONNECT: r.flags |= F_CONNECTED; break;
    case OP_DISCONNECT: r.flags &= ~F_CONNECTED; break;
    case OP_SIDE0: r.flags &= ~F_SIDE; break;
    case OP_SIDE1: r.flags |= F_SIDE; break;
    case OP_CRASH: r.flags &= ~F_CONNECTED; r.flags |= F_DEAD; break;
    case OP_RESTART: r.flags &= ~F_CONNECTED; r.flags |= F_DEAD; make_peer(r, lg, sn, p, c.L, key, now, &sctr); break;
    case OP_GETSTATE: cm.r1 = r.term; cm.r2 = r.isLeader() ? 1 : 0; break;
    }
    r.store(S, gl);
    if (gfl) atomicOr(&S.gflags[g], gfl);
    cmds[i] = cm;
  }
  __syncthreads();
  if (threadIdx.x < RG_N_COUNTERS && sctr.c[threadIdx.x]) atomicAdd((unsigned long long *)&S.counters[threadIdx.x], (unsigned long long)sctr.c[threadIdx.x]);
}

// RequestVote handler, raft.go:200-232, one request per thread (distinct groups per launch)
__global__ void k_rpc_vote(DevPtrs S, DevCfg c, int PP, const int *groups, const int *peers, const int *order, int n,
                           const rg_vote_args *args, rg_vote_reply *replies, int now) {
  __shared__ WaveCtr sctr;
  if (threadIdx.x < RG_N_COUNTERS) sctr.c[threadIdx.x] = 0;
  __syncthreads();
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) {
    const int i = order[q];
    const int g = groups[i], p = peers[i];
    const long long gl = (long long)g * PP + p;
    const Rows lg = rows_of(S.logRows, c, g), sn = rows_of(S.snapRows, c, g);
    int coldLocal[CD_N];
    LaneT<int *> r; r.cold = coldLocal; r.cs = 1; r.load(S, gl);
    if (!r.has(F_DEAD)) {
      rg_vote_args a = args[i];
      VoteDecision d = decide_vote(r, r.cTerm, a.term, a.candidate, a.last_log_index, a.last_log_term);
      commit_vote(r, d, a.term, a.candidate, now, lg, sn, p, c.L, &sctr);
      replies[i].term = d.rTerm; replies[i].granted = d.grant ? 1 : 0;
        r.store(S, gl);
    }
  }
  __syncthreads();
  if (threadIdx.x < RG_N_COUNTERS && sctr.c[threadIdx.x]) atomicAdd((unsigned long long *)&S.counters[threadIdx.x], (unsigned long long)sctr.c[threadIdx.x]);
}

// AppendEntries handler, raft.go:270-321, entries staged in device arrays
__global__ void k_rpc_append(DevPtrs S, DevCfg c, int PP, const int *groups, const int *peers, const int *order, int n,
                             const rg_append_args *args, const int *eterms, const u64 *ecmds, rg_append_reply *replies, int now) {
  __shared__ WaveCtr sctr;
  if (threadIdx.x < RG_N_COUNTERS) sctr.c[threadIdx.x] = 0;
  __syncthreads();
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) {
    const int i = order[q];
    const int g = groups[i], p = peers[i];
    const long long gl = (long long)g * PP + p;
    const Rows lg = rows_of(S.logRows, c, g), sn = rows_of(S.snapRows, c, g);
    int coldLocal[CD_N];
    LaneT<int *> r; r.cold = coldLocal; r.cs = 1; r.load(S, gl);
    if (!r.has(F_DEAD)) {
      rg_append_args a = args[i];
      unsigned gfl = 0;
      int myPrevTerm = a.prev_log_index < r.logLen ? *lg.termp(a.prev_log_index, p) : -2;
      AppendDecision d = decide_append(r, lg, p, a.term, a.prev_log_index, a.prev_log_term, myPrevTerm);
      rg_append_reply rep; rep.term = d.rTerm; rep.success = 0; rep.conflict_term = 0; rep.conflict_first_index = 0;
      if (d.touch) {
        r.setState(RG_FOLLOWER);
        r.vote = a.leader;
        if (r.term < a.term) { r.term = a.term; persist(r, c.L, &sctr); }
        r.timer = now;
        if (d.match) {
          rep.success = 1;
          const int base = a.prev_log_index + 1;
          for (int j = 0; j < a.n_entries; j++) { *lg.termp(base + j, p) = eterms[a.entry_off + j]; *lg.cmdp(base + j, p) = ecmds[a.entry_off + j]; }
          if (a.n_entries > 0 && base < r.snapSync) r.snapSync = base;
          r.logLen = base + a.n_entries;
          r.lastTerm = a.n_entries > 0 ? eterms[a.entry_off + a.n_entries - 1] : a.prev_log_term;
          if (a.leader_commit > r.commit) r.commit = (r.logLen - 1 < a.leader_commit) ? r.logLen - 1 : a.leader_commit;
          if (r.commit > r.logLen - 1) gfl |= RG_FLAG_REF_PANIC;
          __threadfence();
          r.cTerm = *lg.termp(r.commit < c.L ? r.commit : c.L - 1, p);
          persist(r, c.L, &sctr);
          ctr_add(&sctr, RG_C_ENTRIES_REPLICATED, (unsigned)a.n_entries);
        } else {
          rep.conflict_term = d.ct; rep.conflict_first_index = d.ci;
          ctr_add(&sctr, RG_C_AE_CONFLICTS, 1);
        }
      }
      replies[i] = rep;
        r.store(S, gl);
      if (gfl) atomicOr(&S.gflags[g], gfl);
    }
  }
  __syncthreads();
  if (threadIdx.x < RG_N_COUNTERS && sctr.c[threadIdx.x]) atomicAdd((unsigned long long *)&S.counters[threadIdx.x], (unsigned long long)sctr.c[threadIdx.x]);
}

// (re)build the derived words after the host wrote state directly
__global__ void k_derive(DevPtrs S, DevCfg c, int PP, int resetSync) {
  long long gl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int g = (int)(gl / PP), p = (int)(gl % PP);
  if (g >= c.G || p >= c.P) return;
  const Rows lg = rows_of(S.logRows, c, g);
  int len = S.logLen[gl], term = S.term[gl];
  int li = len - 1; if (li < 0) li = 0; if (li >= c.L) li = c.L - 1;
  S.lastTerm[gl] = *lg.termp(li, p);
  { int ci = S.commit[gl]; if (ci < 0) ci = 0; if (ci >= c.L) ci = c.L - 1; S.commitTerm[gl] = *lg.termp(ci, p); }
  int f = len;
  while (f > 0 && f <= c.L && *lg.termp(f - 1, p) == term) f--;
  S.firstOwn[gl] = f;
  S.flags[gl] |= (int)F_CDIRTY;
  if (resetSync) S.snapSync[gl] = 0;   // log and snapshot rows are no longer known equal
}

// Gives every peer whose persisted log still shares the log rows a private copy in the snapshot rows
// (before the host overwrites log rows directly).
__global__ void k_materialize_snap(DevPtrs S, DevCfg c, int PP) {
  long long gl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int g = (int)(gl / PP), p = (int)(gl % PP);
  if (g >= c.G || p >= c.P) return;
  unsigned f = (unsigned)S.flags[gl];
  if (!(f & F_SNAPHAS) || (f & F_SNAPEXT)) return;
  const Rows lg = rows_of(S.logRows, c, g), sn = rows_of(S.snapRows, c, g);
  int n = S.snapLen[gl] < c.L ? S.snapLen[gl] : c.L;
  for (int i = 0; i < n; i++) { *sn.termp(i, p) = *lg.termp(i, p); *sn.cmdp(i, p) = *lg.cmdp(i, p); }
  S.flags[gl] = (int)(f | F_SNAPEXT);
}

// Log matching and leader completeness, one thread per (group, index)
__global__ void k_check(DevPtrs S, DevCfg c, int PP) {
  long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int g = (int)(x / c.L), i = (int)(x % c.L);
  if (g >= c.G || i < 1) return;
  const Rows lg = rows_of(S.logRows, c, g);
  unsigned fl = 0;
  int len[RG_MAX_PEERS]; int tm[RG_MAX_PEERS], tp[RG_MAX_PEERS]; u64 cm[RG_MAX_PEERS]; bool lead[RG_MAX_PEERS];
  bool any = false;
  for (int p = 0; p < c.P; p++) {
    long long gl = (long long)g * PP + p;
    len[p] = S.logLen[gl];
    unsigned f = (unsigned)S.flags[gl];
    lead[p] = (f & (F_DEAD | F_STOP | F_STATE_MASK)) == (unsigned)RG_LEADER;
    if (i < len[p]) { tm[p] = *lg.termp(i, p); tp[p] = *lg.termp(i - 1, p); cm[p] = *lg.cmdp(i, p); any = true; }
    else { tm[p] = 0; tp[p] = 0; cm[p] = 0; }
  }
  if (any)
    for (int a = 0; a < c.P; a++) for (int b = a + 1; b < c.P; b++)
      if (i < len[a] && i < len[b] && tm[a] == tm[b] && (cm[a] != cm[b] || tp[a] != tp[b])) fl |= RG_FLAG_LOG_MATCHING;
  if (i <= S.recMax[g]) {
    u64 rv = S.applyRec[(size_t)g * c.L + i];
    for (int a = 0; a < c.P; a++) if (lead[a] && (i >= len[a] || cm[a] != rv)) fl |= RG_FLAG_LEADER_COMPLETENESS;
  }
  if (fl) atomicOr(&S.gflags[g], fl);
}

__global__ void k_flag_summary(DevPtrs S, int G) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned f = g < G ? S.gflags[g] : 0u;
  if (f) { atomicOr(&S.flagSummary[0], f); atomicAdd(&S.flagSummary[1], 1u); }
}

// state digest: order-free sum of per-word hashes, one lane per (group, peer)
template <int PP>
__global__ __launch_bounds__(256) void k_hash(DevPtrs S, DevCfg c, u64 *out) {
  const Grp<PP> G;
  const long long gl = (long long)blockIdx.x * 256 + threadIdx.x;
  const int g = (int)(gl / PP), p = G.peer;
  const bool valid = g < c.G && p < c.P;
  u64 s = 0;
  if (valid) {
    const Rows lg = rows_of(S.logRows, c, g), sn = rows_of(S.snapRows, c, g);
    unsigned f = (unsigned)S.flags[gl];
    int v[RG_F_PEER_I32_END];
    v[RG_F_TERM] = S.term[gl]; v[RG_F_VOTE] = S.vote[gl]; v[RG_F_STATE] = (int)(f & F_STATE_MASK); v[RG_F_COMMIT] = S.commit[gl];
    v[RG_F_APPLIED] = S.applied[gl]; v[RG_F_LOGLEN] = S.logLen[gl]; v[RG_F_TIMER] = S.timer[gl]; v[RG_F_TIMEOUT] = S.timeout[gl];
    v[RG_F_HBTIMER] = S.hbTimer[gl]; v[RG_F_STOP] = (f & F_STOP) ? 1 : 0; v[RG_F_CNT] = (int)((f & F_CNT_MASK) >> F_CNT_SHIFT);
    v[RG_F_DEAD] = (f & F_DEAD) ? 1 : 0; v[RG_F_RESTART] = (f & F_RESTART) ? 1 : 0; v[RG_F_FIRST] = (f & F_FIRST) ? 1 : 0;
    v[RG_F_MONALIVE] = (f & F_MONALIVE) ? 1 : 0; v[RG_F_ELNEXT] = S.elNext[gl]; v[RG_F_MONNEXT] = S.monNext[gl];
    v[RG_F_RNGCTR] = S.rngCtr[gl]; v[RG_F_ROUNDSEQ] = S.roundSeq[gl]; v[RG_F_SNAPHAS] = (f & F_SNAPHAS) ? 1 : 0;
    v[RG_F_SNAPTERM] = S.snapTerm[gl]; v[RG_F_SNAPVOTE] = S.snapVote[gl]; v[RG_F_SNAPLEN] = S.snapLen[gl];
    v[RG_F_CONNECTED] = (f & F_CONNECTED) ? 1 : 0; v[RG_F_SIDE] = (f & F_SIDE) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < RG_F_PEER_I32_END; k++) s += elem_hash((unsigned)p, (unsigned)k, 0, (u64)(unsigned)v[k]);
    s += elem_hash((unsigned)p, RG_F_APPLYHASH, 0, S.applyHash[gl]);
    for (int i = 0; i < c.P; i++) {
      s += elem_hash((unsigned)p, RG_F_NEXTINDEX, (unsigned)i, (u64)(unsigned)S.nextIdx[((size_t)g * PP + p) * PP + i]);
      s += elem_hash((unsigned)p, RG_F_MATCHINDEX, (unsigned)i, (u64)(unsigned)S.matchIdx[((size_t)g * PP + p) * PP + i]);
    }
    int n = v[RG_F_LOGLEN] < c.L ? v[RG_F_LOGLEN] : c.L;
    for (int i = 0; i < n; i++) {
      s += elem_hash((unsigned)p, RG_F_LOGTERM, (unsigned)i, (u64)(unsigned)*lg.termp(i, p));
      s += elem_hash((unsigned)p, RG_F_LOGCMD, (unsigned)i, *lg.cmdp(i, p));
    }
    n = v[RG_F_SNAPLEN] < c.L ? v[RG_F_SNAPLEN] : c.L;
    const Rows &ps = (f & F_SNAPEXT) ? sn : lg;          // where the persisted log lives
    for (int i = 0; i < n; i++) {
      s += elem_hash((unsigned)p, RG_F_SNAPLOGTERM, (unsigned)i, (u64)(unsigned)*ps.termp(i, p));
      s += elem_hash((unsigned)p, RG_F_SNAPLOGCMD, (unsigned)i, *ps.cmdp(i, p));
    }
    if (p == 0) {
      s += elem_hash(255, RG_F_STARTSEQ, 0, (u64)(unsigned)S.startSeq[g]);
      s += elem_hash(255, RG_F_PARTNEXT, 0, (u64)(unsigned)S.partNext[g]);
      s += elem_hash(255, RG_F_PARTEPOCH, 0, (u64)(unsigned)S.partEpoch[g]);
      s += elem_hash(255, RG_F_RECMAX, 0, (u64)(unsigned)S.recMax[g]);
      s += elem_hash(255, RG_F_GFLAGS, 0, (u64)S.gflags[g]);
      int rm = S.recMax[g];
      for (int i = 1; i <= rm && i < c.L; i++) s += elem_hash(255, RG_F_APPLYREC, (unsigned)i, S.applyRec[(size_t)g * c.L + i]);
    }
  }
#pragma unroll
  for (int m = 1; m < PP; m <<= 1) {
    unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute((G.lane ^ m) << 2, (int)(unsigned)s);
    unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((G.lane ^ m) << 2, (int)(unsigned)(s >> 32));
    s += ((u64)hi << 32) | lo;
  }
  if (valid && p == 0) out[g] = sm64(s ^ (u64)(c.groupBase + g));
}

// ---------------------------------------------------------------- host side
struct rg_engine {
  rg_config cfg;
  int G, P, L, PP, Gpad, gpb, rowBytes, cmdOff;
  long long nLanes;
  DevPtrs S;
  DevPtrs *dS;          // device copy of S: the tick kernel reads the cold pointers through it
  HotPtrs H;
  DevCfg dc;
  std::vector<void *> allocs;
  hipStream_t ownStream, stream;
  hipEvent_t ev0, ev1;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pendingEv;
  double stepMs;
  long long launches;
  long long now;
  bool derivedDirty, syncDirty;
  std::vector<int> drained;       // per (g,p): last index handed out by rg_drain_applied
  std::mutex mu;
  // staging
  HostCmd *dCmd; size_t dCmdCap;
};

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { return RG_ERR_HIP; } } while (0)

static const char *kVersion = "raftgpu 0.1 (gfx950)";

extern "C" const char *rg_version(void) { return kVersion; }
extern "C" const char *rg_strerror(int st) {
  switch (st) {
  case RG_OK: return "ok";
  case RG_ERR_ARG: return "bad argument";
  case RG_ERR_HIP: return "HIP runtime or device failure";
  case RG_ERR_NOMEM: return "out of memory";
  case RG_ERR_RANGE: return "group/peer/index out of range";
  case RG_ERR_STATE: return "illegal in this state";
  default: return "unknown status";
  }
}

extern "C" int rg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return RG_ERR_HIP;
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

static bool valid_config(const rg_config *c) {
  if (!c) return false;
  if (c->n_groups < 1 || c->n_peers < 2 || c->n_peers > RG_MAX_PEERS || c->log_cap < 2) return false;
  if (c->start_period < 0 || c->part_period < 0 || c->churn_period < 0) return false;
  if (c->start_period > 0 && (c->start_max < 0 || c->start_max + 1 > c->log_cap)) return false;
  if (c->req_drop_permille < 0 || c->req_drop_permille > 1000 || c->reply_drop_permille < 0 || c->reply_drop_permille > 1000 ||
      c->reply_delay_permille < 0 || c->reply_delay_permille > 1000) return false;
  return true;
}

template <typename T>
static int dalloc(rg_engine *e, T **p, size_t n) {
  void *q = nullptr;
  if (hipMalloc(&q, n * sizeof(T)) != hipSuccess) return RG_ERR_NOMEM;
  e->allocs.push_back(q);
  *p = (T *)q;
  return RG_OK;
}

static void fill_devcfg(rg_engine *e) {
  DevCfg &d = e->dc; const rg_config &c = e->cfg;
  d.G = e->G; d.P = e->P; d.L = e->L; d.rowBytes = e->rowBytes; d.cmdOff = e->cmdOff;
  d.groupBase = c.group_base; d.seedKey = sm64(c.seed);
  d.reqDrop = c.req_drop_permille; d.replyDrop = c.reply_drop_permille; d.replyDelay = c.reply_delay_permille;
  d.startPeriod = c.start_period; d.startFirst = c.start_first_tick; d.startMax = c.start_max; d.startMode = c.start_mode;
  d.partPeriod = c.part_period; d.partJitter = c.part_jitter;
  d.churnPeriod = c.churn_period; d.churnDisc = c.churn_disconnect_permille; d.churnRestart = c.churn_restart_permille; d.churnCrash = c.churn_crash_permille;
  d.acctBytes = c.account_bytes != 0;
}

// Make for every peer: device memsets plus a host-built image of the few non-zero words.
static int init_state(rg_engine *e) {
  const long long nL = e->nLanes;
  DevPtrs &S = e->S;
  hipStream_t st = e->stream;
  int *zeroed[] = {S.term, S.commit, S.applied, S.timer, S.hbTimer, S.elNext, S.monNext, S.roundSeq, S.snapTerm, S.snapVote, S.snapLen, S.lastTerm, S.commitTerm};
  for (int *p : zeroed) HIPCHK(hipMemsetAsync(p, 0, nL * 4, st));
  HIPCHK(hipMemsetAsync(S.applyHash, 0, nL * 8, st));
  HIPCHK(hipMemsetAsync(S.vote, 0xFF, nL * 4, st));            // -1
  HIPCHK(hipMemsetAsync(S.nextIdx, 0, (size_t)e->Gpad * e->PP * e->PP * 4, st));
  HIPCHK(hipMemsetAsync(S.matchIdx, 0, (size_t)e->Gpad * e->PP * e->PP * 4, st));
  HIPCHK(hipMemsetAsync(S.logRows, 0, (size_t)e->G * e->L * e->rowBytes, st));
  if (S.snapRows) HIPCHK(hipMemsetAsync(S.snapRows, 0, (size_t)e->G * e->L * e->rowBytes, st));
  HIPCHK(hipMemsetAsync(S.applyRec, 0, (size_t)e->G * e->L * 8, st));
  HIPCHK(hipMemsetAsync(S.startSeq, 0, (size_t)e->Gpad * 4, st));
  HIPCHK(hipMemsetAsync(S.partEpoch, 0, (size_t)e->Gpad * 4, st));
  HIPCHK(hipMemsetAsync(S.recMax, 0, (size_t)e->Gpad * 4, st));
  HIPCHK(hipMemsetAsync(S.gflags, 0, (size_t)e->Gpad * 4, st));
  HIPCHK(hipMemsetAsync(S.counters, 0, RG_CTR_SLOTS * 8, st));
  HIPCHK(hipMemsetAsync(S.flagSummary, 0, 8, st));
  // words that depend on (group, peer): flags, logLen, snapSync, firstOwn, timeout, rngCtr, partNext
  std::vector<int> flags(nL, 0), one(nL, 0), timeout(nL, 0), rng(nL, 0), pn(e->Gpad, 0);
  const u64 seedKey = sm64(e->cfg.seed);
  for (int g = 0; g < e->G; g++) {
    const u64 key = sm64(seedKey ^ (u64)(e->cfg.group_base + g));
    pn[g] = e->cfg.part_period > 0 ? e->cfg.part_period : 0;
    for (int p = 0; p < e->P; p++) {
      long long gl = (long long)g * e->PP + p;
      flags[gl] = (int)F_CONNECTED;
      one[gl] = 1;
      if (p != 0) {                                              // raft.go:545-547
        u64 h = stream_hash(key, STREAM_ELECTION, (unsigned)p, 0);
        timeout[gl] = (int)((unsigned)(h >> 32) % (unsigned)RG_TIMEOUT_SPAN_MS) + RG_TIMEOUT_BASE_MS;
        rng[gl] = 1;
      }
    }
  }
  HIPCHK(hipMemcpyAsync(S.flags, flags.data(), nL * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S.logLen, one.data(), nL * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S.snapSync, one.data(), nL * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S.firstOwn, one.data(), nL * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S.timeout, timeout.data(), nL * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S.rngCtr, rng.data(), nL * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S.partNext, pn.data(), (size_t)e->Gpad * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  e->now = 0; e->derivedDirty = false; e->syncDirty = false;
  std::fill(e->drained.begin(), e->drained.end(), 0);
  return RG_OK;
}

static void free_engine(rg_engine *e) {
  for (auto &pr : e->pendingEv) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  for (void *p : e->allocs) hipFree(p);
  if (e->dCmd) hipFree(e->dCmd);
  if (e->ownStream) hipStreamDestroy(e->ownStream);
  delete e;
}

// The separate snapshot rows exist only once something needs them: an imported Persister blob or a
// host write to the log arrays.
static int ensure_snap_rows(rg_engine *e) {
  if (e->S.snapRows) return RG_OK;
  const size_t n = (size_t)e->G * e->L * e->rowBytes;
  int rc = dalloc(e, &e->S.snapRows, n);
  if (rc != RG_OK) return rc;
  HIPCHK(hipMemsetAsync(e->S.snapRows, 0, n, e->stream));
  e->H.snapRows = e->S.snapRows;
  HIPCHK(hipMemcpyAsync(e->dS, &e->S, sizeof(DevPtrs), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_create(const rg_config *cfg, rg_engine **out) {
  if (!out || !valid_config(cfg)) return RG_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return RG_ERR_HIP;
  if (cfg->device < 0 || cfg->device >= ndev) return RG_ERR_ARG;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return RG_ERR_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RG_ERR_HIP;   // kernels are built for gfx950 only
  HIPCHK(hipSetDevice(cfg->device));
  rg_engine *e = new (std::nothrow) rg_engine();
  if (!e) return RG_ERR_NOMEM;
  e->cfg = *cfg; e->G = cfg->n_groups; e->P = cfg->n_peers; e->L = cfg->log_cap;
  e->PP = e->P <= 4 ? 4 : 8;
  e->gpb = 4 * (64 / e->P);          // groups per block of the tick kernel: 64/P whole groups per wave
  e->Gpad = (e->G + e->gpb - 1) / e->gpb * e->gpb;
  e->nLanes = (long long)e->Gpad * e->PP;
  e->cmdOff = (e->P * 4 + 7) & ~7;
  e->rowBytes = (e->cmdOff + e->P * 8 + 15) & ~15;
  e->ownStream = nullptr; e->dCmd = nullptr; e->dCmdCap = 0; e->stepMs = 0; e->launches = 0;
  if (hipStreamCreateWithFlags(&e->ownStream, hipStreamNonBlocking) != hipSuccess) { delete e; return RG_ERR_HIP; }
  e->stream = e->ownStream;
  DevPtrs &S = e->S;
  int rc = RG_OK;
  int **i32s[] = {&S.term, &S.vote, &S.flags, &S.commit, &S.applied, &S.logLen, &S.timer, &S.timeout, &S.hbTimer, &S.elNext, &S.monNext,
                  &S.rngCtr, &S.roundSeq, &S.snapTerm, &S.snapVote, &S.snapLen, &S.snapSync, &S.lastTerm, &S.firstOwn, &S.commitTerm};
  for (int **p : i32s) if (rc == RG_OK) rc = dalloc(e, p, (size_t)e->nLanes);
  if (rc == RG_OK) rc = dalloc(e, &S.applyHash, (size_t)e->nLanes);
  if (rc == RG_OK) rc = dalloc(e, &S.nextIdx, (size_t)e->Gpad * e->PP * e->PP);
  if (rc == RG_OK) rc = dalloc(e, &S.matchIdx, (size_t)e->Gpad * e->PP * e->PP);
  if (rc == RG_OK) rc = dalloc(e, &S.logRows, (size_t)e->G * e->L * e->rowBytes);
  S.snapRows = nullptr;     // allocated when a blob is first imported (ensure_snap_rows)
  if (rc == RG_OK) rc = dalloc(e, &S.applyRec, (size_t)e->G * e->L);
  if (rc == RG_OK) rc = dalloc(e, &S.startSeq, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.partNext, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.partEpoch, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.recMax, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.gflags, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.counters, (size_t)RG_CTR_SLOTS);
  if (rc == RG_OK) rc = dalloc(e, &S.flagSummary, (size_t)2);
  if (rc != RG_OK) { free_engine(e); return rc; }
  e->drained.assign((size_t)e->G * e->P, 0);
  fill_devcfg(e);
  e->H.logRows = S.logRows; e->H.snapRows = S.snapRows; e->H.nextIdx = S.nextIdx; e->H.matchIdx = S.matchIdx; e->H.applyRec = S.applyRec;
  if (dalloc(e, &e->dS, 1) != RG_OK || hipMemcpy(e->dS, &e->S, sizeof(DevPtrs), hipMemcpyHostToDevice) != hipSuccess) { free_engine(e); return RG_ERR_NOMEM; }
  rc = init_state(e);
  if (rc != RG_OK) { free_engine(e); return rc; }
  *out = e;
  return RG_OK;
}

extern "C" int rg_destroy(rg_engine *e) {
  if (!e) return RG_ERR_ARG;
  hipSetDevice(e->cfg.device);
  hipStreamSynchronize(e->stream);
  free_engine(e);
  return RG_OK;
}

#define LOCK(e) if (!(e)) return RG_ERR_ARG; std::lock_guard<std::mutex> _lk((e)->mu); if (hipSetDevice((e)->cfg.device) != hipSuccess) return RG_ERR_HIP

static int collect_events(rg_engine *e) {
  for (auto &pr : e->pendingEv) {
    float ms = 0;
    if (hipEventSynchronize(pr.second) != hipSuccess) return RG_ERR_HIP;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) != hipSuccess) return RG_ERR_HIP;
    e->stepMs += ms;
    hipEventDestroy(pr.first); hipEventDestroy(pr.second);
  }
  e->pendingEv.clear();
  return RG_OK;
}

extern "C" int rg_reset(rg_engine *e) {
  LOCK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  int rc = collect_events(e);
  if (rc != RG_OK) return rc;
  return init_state(e);
}

extern "C" int rg_set_stream(rg_engine *e, void *s) {
  LOCK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  e->stream = s ? (hipStream_t)s : e->ownStream;
  return RG_OK;
}

static int run_derive(rg_engine *e) {
  if (!e->derivedDirty) return RG_OK;
  int blocks = (int)((e->nLanes + 255) / 256);
  hipLaunchKernelGGL(k_derive, dim3(blocks), dim3(256), 0, e->stream, e->S, e->dc, e->PP, e->syncDirty ? 1 : 0);
  HIPCHK(hipGetLastError());
  e->derivedDirty = false; e->syncDirty = false;
  return RG_OK;
}

// The per-wave counters of a launch are 32-bit words in LDS; the fastest of them (wire bytes) can grow by a few
// thousand per tick, so one launch covers at most this many ticks.
#define RG_MAX_LAUNCH_TICKS 500000

static int launch_ticks(rg_engine *e, int n_ticks) {
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  const int blocks = e->Gpad / e->gpb;
  HIPCHK(hipEventRecord(a, e->stream));
  const DevPtrs *dS = e->dS;
#define TICK_LAUNCH(PEERS) do { \
    if (e->dc.acctBytes) hipLaunchKernelGGL((k_tick<PEERS, true>), dim3(blocks), dim3(256), 0, e->stream, dS, e->H, e->dc, (int)e->now, n_ticks); \
    else hipLaunchKernelGGL((k_tick<PEERS, false>), dim3(blocks), dim3(256), 0, e->stream, dS, e->H, e->dc, (int)e->now, n_ticks); } while (0)
  switch (e->P) {
  case 2: TICK_LAUNCH(2); break;
  case 3: TICK_LAUNCH(3); break;
  case 4: TICK_LAUNCH(4); break;
  case 5: TICK_LAUNCH(5); break;
  case 6: TICK_LAUNCH(6); break;
  case 7: TICK_LAUNCH(7); break;
  default: TICK_LAUNCH(8); break;
  }
#undef TICK_LAUNCH
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(b, e->stream));
  e->pendingEv.push_back(std::make_pair(a, b));
  e->launches++;
  e->now += n_ticks;
  // group-ticks are counted on the host side of the counter block
  return RG_OK;
}

extern "C" int rg_step(rg_engine *e, int32_t n_ticks) {
  LOCK(e);
  if (n_ticks < 0) return RG_ERR_ARG;
  if (n_ticks == 0) return RG_OK;
  if (e->now + n_ticks > 0x7ffffff0ll) return RG_ERR_RANGE;
  int rc = run_derive(e);
  if (rc != RG_OK) return rc;
  while (n_ticks > 0) {
    if (e->pendingEv.size() >= 512) { rc = collect_events(e); if (rc != RG_OK) return rc; }
    const int n = n_ticks < RG_MAX_LAUNCH_TICKS ? n_ticks : RG_MAX_LAUNCH_TICKS;
    rc = launch_ticks(e, n);
    if (rc != RG_OK) return rc;
    n_ticks -= n;
  }
  return RG_OK;
}

extern "C" int rg_sync(rg_engine *e) {
  LOCK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_now(rg_engine *e, int64_t *tick) {
  LOCK(e);
  if (!tick) return RG_ERR_ARG;
  *tick = e->now;
  return RG_OK;
}

extern "C" int rg_step_time_ms(rg_engine *e, double *total_ms, int64_t *launches) {
  LOCK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  int rc = collect_events(e);
  if (rc != RG_OK) return rc;
  if (total_ms) *total_ms = e->stepMs;
  if (launches) *launches = e->launches;
  return RG_OK;
}

static int ensure_cmd_buf(rg_engine *e, size_t n) {
  if (n <= e->dCmdCap) return RG_OK;
  if (e->dCmd) { hipFree(e->dCmd); e->dCmd = nullptr; e->dCmdCap = 0; }
  size_t cap = std::max<size_t>(n, 1024);
  if (hipMalloc((void **)&e->dCmd, cap * sizeof(HostCmd)) != hipSuccess) return RG_ERR_NOMEM;
  e->dCmdCap = cap;
  return RG_OK;
}

// Runs host commands in order.  Commands on one group are kept in order by
// cutting the batch into launches in which every group appears at most once.
static int run_cmds(rg_engine *e, std::vector<HostCmd> &cmds) {
  const size_t n = cmds.size();
  if (n == 0) return RG_OK;
  for (auto &cm : cmds) if (cm.group < 0 || cm.group >= e->G || cm.peer < 0 || cm.peer >= e->P) return RG_ERR_RANGE;
  int rc = run_derive(e);
  if (rc != RG_OK) return rc;
  std::vector<int> level(n, 0);
  int maxLevel = 0;
  if (n > 1) {
    std::unordered_map<int, int> last;
    for (size_t i = 0; i < n; i++) {
      auto it = last.find(cmds[i].group);
      int lv = it == last.end() ? 0 : it->second + 1;
      last[cmds[i].group] = lv; level[i] = lv; maxLevel = std::max(maxLevel, lv);
    }
  }
  rc = ensure_cmd_buf(e, n);
  if (rc != RG_OK) return rc;
  std::vector<HostCmd> sorted; sorted.reserve(n);
  std::vector<size_t> src; src.reserve(n);
  std::vector<size_t> startOf(maxLevel + 2, 0);
  for (int lv = 0; lv <= maxLevel; lv++) {
    startOf[lv] = sorted.size();
    for (size_t i = 0; i < n; i++) if (level[i] == lv) { sorted.push_back(cmds[i]); src.push_back(i); }
  }
  startOf[maxLevel + 1] = sorted.size();
  HIPCHK(hipMemcpyAsync(e->dCmd, sorted.data(), n * sizeof(HostCmd), hipMemcpyHostToDevice, e->stream));
  for (int lv = 0; lv <= maxLevel; lv++) {
    int cnt = (int)(startOf[lv + 1] - startOf[lv]);
    if (cnt == 0) continue;
    hipLaunchKernelGGL(k_host_cmds, dim3((cnt + 127) / 128), dim3(128), 0, e->stream, e->S, e->dc, e->PP, e->dCmd + startOf[lv], cnt, (int)e->now * RG_TICK_MS);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(sorted.data(), e->dCmd, n * sizeof(HostCmd), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < n; i++) cmds[src[i]] = sorted[i];
  return RG_OK;
}

static int run_one(rg_engine *e, int op, int g, int p, u64 arg, HostCmd *res) {
  std::vector<HostCmd> v(1);
  memset(&v[0], 0, sizeof(HostCmd));
  v[0].op = op; v[0].group = g; v[0].peer = p; v[0].arg = arg;
  int rc = run_cmds(e, v);
  if (rc == RG_OK && res) *res = v[0];
  return rc;
}

extern "C" int rg_start(rg_engine *e, int32_t g, int32_t p, uint64_t cmd, int32_t *index, int32_t *term, int32_t *is_leader) {
  LOCK(e);
  HostCmd r;
  int rc = run_one(e, OP_START, g, p, cmd, &r);
  if (rc != RG_OK) return rc;
  if (index) *index = r.r0;
  if (term) *term = r.r1;
  if (is_leader) *is_leader = r.r2;
  return RG_OK;
}

extern "C" int rg_start_batch(rg_engine *e, rg_start_req *reqs, int64_t n) {
  LOCK(e);
  if (n < 0 || (n > 0 && !reqs)) return RG_ERR_ARG;
  std::vector<HostCmd> v((size_t)n);
  for (int64_t i = 0; i < n; i++) { memset(&v[i], 0, sizeof(HostCmd)); v[i].op = OP_START; v[i].group = reqs[i].group; v[i].peer = reqs[i].peer; v[i].arg = reqs[i].command; }
  int rc = run_cmds(e, v);
  if (rc != RG_OK) return rc;
  for (int64_t i = 0; i < n; i++) { reqs[i].index = v[i].r0; reqs[i].term = v[i].r1; reqs[i].is_leader = v[i].r2; }
  return RG_OK;
}

extern "C" int rg_get_state(rg_engine *e, int32_t g, int32_t p, int32_t *term, int32_t *is_leader) {
  LOCK(e);
  HostCmd r;
  int rc = run_one(e, OP_GETSTATE, g, p, 0, &r);
  if (rc != RG_OK) return rc;
  if (term) *term = r.r1;
  if (is_leader) *is_leader = r.r2;
  return RG_OK;
}

extern "C" int rg_set_connected(rg_engine *e, int32_t g, int32_t p, int32_t on) { LOCK(e); return run_one(e, on ? OP_CONNECT : OP_DISCONNECT, g, p, 0, nullptr); }
extern "C" int rg_set_side(rg_engine *e, int32_t g, int32_t p, int32_t side) { LOCK(e); return run_one(e, (side & 1) ? OP_SIDE1 : OP_SIDE0, g, p, 0, nullptr); }
extern "C" int rg_crash(rg_engine *e, int32_t g, int32_t p) { LOCK(e); return run_one(e, OP_CRASH, g, p, 0, nullptr); }
extern "C" int rg_restart(rg_engine *e, int32_t g, int32_t p) {
  LOCK(e);
  int rc = run_one(e, OP_RESTART, g, p, 0, nullptr);
  if (rc == RG_OK) e->drained[(size_t)g * e->P + p] = 0;     // the new instance replays from index 1 (raft.go:717)
  return rc;
}

extern "C" int rg_set_unreliable(rg_engine *e, int32_t a, int32_t b, int32_t d) {
  LOCK(e);
  if (a < 0 || b < 0 || d < 0 || a > 1000 || b > 1000 || d > 1000) return RG_ERR_ARG;
  e->cfg.req_drop_permille = a; e->cfg.reply_drop_permille = b; e->cfg.reply_delay_permille = d;
  fill_devcfg(e);
  return RG_OK;
}

// ---- strided host <-> device copies of one column / word ------------------------
static int d2h(rg_engine *e, void *dst, const void *src, size_t n) {
  HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}
static int h2d(rg_engine *e, void *dst, const void *src, size_t n) {
  HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_drain_applied(rg_engine *e, int32_t g, int32_t p, rg_apply_msg *buf, int64_t cap, int64_t *n_out) {
  LOCK(e);
  if (!n_out || cap < 0 || (cap > 0 && !buf)) return RG_ERR_ARG;
  if (g < 0 || g >= e->G || p < 0 || p >= e->P) return RG_ERR_RANGE;
  int applied = 0;
  int rc = d2h(e, &applied, e->S.applied + (size_t)g * e->PP + p, 4);
  if (rc != RG_OK) return rc;
  int &cur = e->drained[(size_t)g * e->P + p];
  if (applied < cur) cur = 0;
  int64_t n = applied - cur;
  if (n > cap) n = cap;
  if (n > 0) {
    std::vector<char> rows((size_t)n * e->rowBytes);
    rc = d2h(e, rows.data(), e->S.logRows + ((size_t)g * e->L + cur + 1) * e->rowBytes, rows.size());
    if (rc != RG_OK) return rc;
    for (int64_t i = 0; i < n; i++) {
      buf[i].command_index = cur + 1 + i;
      memcpy(&buf[i].command, rows.data() + (size_t)i * e->rowBytes + e->cmdOff + 8 * (size_t)p, 8);
    }
    cur += (int)n;
  }
  *n_out = n;
  return RG_OK;
}

extern "C" int rg_snapshot_export(rg_engine *e, int32_t g, int32_t p, void *buf, int64_t cap, int64_t *size_out) {
  LOCK(e);
  if (g < 0 || g >= e->G || p < 0 || p >= e->P) return RG_ERR_RANGE;
  const size_t gl = (size_t)g * e->PP + p;
  int flags = 0, hdr[3];
  int rc = d2h(e, &flags, e->S.flags + gl, 4);
  if (rc == RG_OK) rc = d2h(e, &hdr[0], e->S.snapTerm + gl, 4);
  if (rc == RG_OK) rc = d2h(e, &hdr[1], e->S.snapVote + gl, 4);
  if (rc == RG_OK) rc = d2h(e, &hdr[2], e->S.snapLen + gl, 4);
  if (rc != RG_OK) return rc;
  const bool has = ((unsigned)flags & F_SNAPHAS) != 0;
  const int n = has ? hdr[2] : 0;
  const int64_t size = has ? 16 + 16 * (int64_t)n : 0;
  if (size_out) *size_out = size;
  if (!buf) return RG_OK;
  if (cap < size) return RG_ERR_ARG;
  if (!has) return RG_OK;
  int32_t *h = (int32_t *)buf;
  h[0] = hdr[0]; h[1] = hdr[1]; h[2] = n; h[3] = 0;
  if (n > 0) {
    std::vector<char> rows((size_t)n * e->rowBytes);
    const char *srcRows = (((unsigned)flags & F_SNAPEXT) && e->S.snapRows) ? e->S.snapRows : e->S.logRows;
    rc = d2h(e, rows.data(), srcRows + (size_t)g * e->L * e->rowBytes, rows.size());
    if (rc != RG_OK) return rc;
    for (int i = 0; i < n; i++) {
      char *o = (char *)buf + 16 + 16 * (size_t)i;
      memcpy(o, rows.data() + (size_t)i * e->rowBytes + 4 * (size_t)p, 4);
      memset(o + 4, 0, 4);
      memcpy(o + 8, rows.data() + (size_t)i * e->rowBytes + e->cmdOff + 8 * (size_t)p, 8);
    }
  }
  return RG_OK;
}

extern "C" int rg_snapshot_import(rg_engine *e, int32_t g, int32_t p, const void *buf, int64_t size) {
  LOCK(e);
  if (g < 0 || g >= e->G || p < 0 || p >= e->P) return RG_ERR_RANGE;
  const size_t gl = (size_t)g * e->PP + p;
  int flags = 0;
  int rc = d2h(e, &flags, e->S.flags + gl, 4);
  if (rc != RG_OK) return rc;
  if (size == 0) {
    flags &= ~(int)F_SNAPHAS;
    int zero = 0;
    rc = h2d(e, e->S.flags + gl, &flags, 4);
    if (rc == RG_OK) rc = h2d(e, e->S.snapLen + gl, &zero, 4);
    return rc;
  }
  if (!buf || size < 16) return RG_ERR_ARG;
  const int32_t *h = (const int32_t *)buf;
  const int n = h[2];
  if (n < 0 || n > e->L || size != 16 + 16 * (int64_t)n) return RG_ERR_ARG;
  rc = ensure_snap_rows(e);
  if (rc != RG_OK) return rc;
  if (n > 0) {
    std::vector<char> rows((size_t)n * e->rowBytes);
    char *dbase = e->S.snapRows + (size_t)g * e->L * e->rowBytes;
    rc = d2h(e, rows.data(), dbase, rows.size());
    if (rc != RG_OK) return rc;
    for (int i = 0; i < n; i++) {
      const char *s = (const char *)buf + 16 + 16 * (size_t)i;
      memcpy(rows.data() + (size_t)i * e->rowBytes + 4 * (size_t)p, s, 4);
      memcpy(rows.data() + (size_t)i * e->rowBytes + e->cmdOff + 8 * (size_t)p, s + 8, 8);
    }
    rc = h2d(e, dbase, rows.data(), rows.size());
    if (rc != RG_OK) return rc;
  }
  flags |= (int)(F_SNAPHAS | F_SNAPEXT);
  int zero = 0;
  rc = h2d(e, e->S.flags + gl, &flags, 4);
  if (rc == RG_OK) rc = h2d(e, e->S.snapTerm + gl, &h[0], 4);
  if (rc == RG_OK) rc = h2d(e, e->S.snapVote + gl, &h[1], 4);
  if (rc == RG_OK) rc = h2d(e, e->S.snapLen + gl, &n, 4);
  if (rc == RG_OK) rc = h2d(e, e->S.snapSync + gl, &zero, 4);
  return rc;
}

// ---- batched RPC handlers --------------------------------------------------------
// order[]: request indices sorted so that each launch level holds one request per group
static void level_order(const int32_t *groups, int64_t n, std::vector<int> &order, std::vector<size_t> &startOf) {
  std::vector<int> level((size_t)n, 0);
  int maxLevel = 0;
  std::unordered_map<int, int> last;
  for (int64_t i = 0; i < n; i++) {
    auto it = last.find(groups[i]);
    int lv = it == last.end() ? 0 : it->second + 1;
    last[groups[i]] = lv; level[i] = lv; maxLevel = std::max(maxLevel, lv);
  }
  order.clear(); startOf.assign(maxLevel + 2, 0);
  for (int lv = 0; lv <= maxLevel; lv++) {
    startOf[lv] = order.size();
    for (int64_t i = 0; i < n; i++) if (level[i] == lv) order.push_back((int)i);
  }
  startOf[maxLevel + 1] = order.size();
}

struct TmpBuf {
  std::vector<void *> p;
  ~TmpBuf() { for (void *q : p) hipFree(q); }
  template <typename T> T *get(size_t n) { void *q = nullptr; if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr; p.push_back(q); return (T *)q; }
};

extern "C" int rg_rpc_request_vote(rg_engine *e, int64_t n, const int32_t *groups, const int32_t *peers,
                                   const rg_vote_args *args, rg_vote_reply *replies) {
  LOCK(e);
  if (n < 0 || (n > 0 && (!groups || !peers || !args || !replies))) return RG_ERR_ARG;
  if (n == 0) return RG_OK;
  for (int64_t i = 0; i < n; i++) if (groups[i] < 0 || groups[i] >= e->G || peers[i] < 0 || peers[i] >= e->P) return RG_ERR_RANGE;
  int rc = run_derive(e);
  if (rc != RG_OK) return rc;
  std::vector<int> order; std::vector<size_t> startOf;
  level_order(groups, n, order, startOf);
  TmpBuf tb;
  int *dg = tb.get<int>(n), *dp = tb.get<int>(n), *dord = tb.get<int>(n);
  rg_vote_args *da = tb.get<rg_vote_args>(n); rg_vote_reply *dr = tb.get<rg_vote_reply>(n);
  if (!dg || !dp || !dord || !da || !dr) return RG_ERR_NOMEM;
  HIPCHK(hipMemcpyAsync(dg, groups, n * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(dp, peers, n * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(dord, order.data(), n * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(da, args, n * sizeof(rg_vote_args), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(dr, replies, n * sizeof(rg_vote_reply), hipMemcpyHostToDevice, e->stream));
  for (size_t lv = 0; lv + 1 < startOf.size(); lv++) {
    int cnt = (int)(startOf[lv + 1] - startOf[lv]);
    if (!cnt) continue;
    hipLaunchKernelGGL(k_rpc_vote, dim3((cnt + 127) / 128), dim3(128), 0, e->stream, e->S, e->dc, e->PP, dg, dp, dord + startOf[lv], cnt, da, dr, (int)e->now * RG_TICK_MS);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(replies, dr, n * sizeof(rg_vote_reply), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_rpc_append_entries(rg_engine *e, int64_t n, const int32_t *groups, const int32_t *peers,
                                     const rg_append_args *args, const int32_t *eterms, const uint64_t *ecmds,
                                     int64_t ne, rg_append_reply *replies) {
  LOCK(e);
  if (n < 0 || ne < 0 || (n > 0 && (!groups || !peers || !args || !replies)) || (ne > 0 && (!eterms || !ecmds))) return RG_ERR_ARG;
  if (n == 0) return RG_OK;
  for (int64_t i = 0; i < n; i++) {
    if (groups[i] < 0 || groups[i] >= e->G || peers[i] < 0 || peers[i] >= e->P) return RG_ERR_RANGE;
    const rg_append_args &a = args[i];
    if (a.prev_log_index < 0 || a.n_entries < 0 || (int64_t)a.prev_log_index + 1 + a.n_entries > e->L) return RG_ERR_RANGE;
    if (a.entry_off < 0 || a.entry_off + a.n_entries > ne) return RG_ERR_RANGE;
  }
  int rc = run_derive(e);
  if (rc != RG_OK) return rc;
  std::vector<int> order; std::vector<size_t> startOf;
  level_order(groups, n, order, startOf);
  TmpBuf tb;
  int *dg = tb.get<int>(n), *dp = tb.get<int>(n), *dord = tb.get<int>(n), *dt = tb.get<int>(ne);
  u64 *dcm = tb.get<u64>(ne);
  rg_append_args *da = tb.get<rg_append_args>(n); rg_append_reply *dr = tb.get<rg_append_reply>(n);
  if (!dg || !dp || !dord || !dt || !dcm || !da || !dr) return RG_ERR_NOMEM;
  HIPCHK(hipMemcpyAsync(dg, groups, n * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(dp, peers, n * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(dord, order.data(), n * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(da, args, n * sizeof(rg_append_args), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(dr, replies, n * sizeof(rg_append_reply), hipMemcpyHostToDevice, e->stream));
  if (ne > 0) {
    HIPCHK(hipMemcpyAsync(dt, eterms, ne * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(dcm, ecmds, ne * 8, hipMemcpyHostToDevice, e->stream));
  }
  for (size_t lv = 0; lv + 1 < startOf.size(); lv++) {
    int cnt = (int)(startOf[lv + 1] - startOf[lv]);
    if (!cnt) continue;
    hipLaunchKernelGGL(k_rpc_append, dim3((cnt + 127) / 128), dim3(128), 0, e->stream, e->S, e->dc, e->PP, dg, dp, dord + startOf[lv], cnt, da, dt, dcm, dr, (int)e->now * RG_TICK_MS);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(replies, dr, n * sizeof(rg_append_reply), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

// ---- bulk state access -----------------------------------------------------------
static int64_t field_bytes(const rg_engine *e, int field) {
  int64_t G = e->G, P = e->P, L = e->L;
  if (field >= 0 && field < RG_F_PEER_I32_END) return G * P * 4;
  switch (field) {
  case RG_F_APPLYHASH: return G * P * 8;
  case RG_F_NEXTINDEX: case RG_F_MATCHINDEX: return G * P * P * 4;
  case RG_F_LOGTERM: case RG_F_SNAPLOGTERM: return G * P * L * 4;
  case RG_F_LOGCMD: case RG_F_SNAPLOGCMD: return G * P * L * 8;
  case RG_F_STARTSEQ: case RG_F_PARTNEXT: case RG_F_PARTEPOCH: case RG_F_RECMAX: case RG_F_GFLAGS: return G * 4;
  case RG_F_APPLYREC: return G * L * 8;
  default: return -1;
  }
}

static unsigned flag_bit_of(int field, int *shift, unsigned *mask) {
  *shift = 0; *mask = 1;
  switch (field) {
  case RG_F_STATE: *mask = 3; return 1;
  case RG_F_STOP: *shift = 2; return 1;
  case RG_F_DEAD: *shift = 3; return 1;
  case RG_F_RESTART: *shift = 4; return 1;
  case RG_F_FIRST: *shift = 5; return 1;
  case RG_F_MONALIVE: *shift = 6; return 1;
  case RG_F_SNAPHAS: *shift = 7; return 1;
  case RG_F_CONNECTED: *shift = 8; return 1;
  case RG_F_SIDE: *shift = 9; return 1;
  case RG_F_CNT: *shift = F_CNT_SHIFT; *mask = 7; return 1;
  default: return 0;
  }
}

static int *peer_word(rg_engine *e, int field) {
  DevPtrs &S = e->S;
  switch (field) {
  case RG_F_TERM: return S.term; case RG_F_VOTE: return S.vote; case RG_F_COMMIT: return S.commit; case RG_F_APPLIED: return S.applied;
  case RG_F_LOGLEN: return S.logLen; case RG_F_TIMER: return S.timer; case RG_F_TIMEOUT: return S.timeout; case RG_F_HBTIMER: return S.hbTimer;
  case RG_F_ELNEXT: return S.elNext; case RG_F_MONNEXT: return S.monNext; case RG_F_RNGCTR: return S.rngCtr; case RG_F_ROUNDSEQ: return S.roundSeq;
  case RG_F_SNAPTERM: return S.snapTerm; case RG_F_SNAPVOTE: return S.snapVote; case RG_F_SNAPLEN: return S.snapLen;
  default: return nullptr;
  }
}
static int *group_word(rg_engine *e, int field) {
  DevPtrs &S = e->S;
  switch (field) {
  case RG_F_STARTSEQ: return S.startSeq; case RG_F_PARTNEXT: return S.partNext; case RG_F_PARTEPOCH: return S.partEpoch;
  case RG_F_RECMAX: return S.recMax; case RG_F_GFLAGS: return (int *)S.gflags;
  default: return nullptr;
  }
}

static int rw_state(rg_engine *e, int field, void *buf, int64_t bytes, bool wr) {
  if (!buf) return RG_ERR_ARG;
  const int64_t need = field_bytes(e, field);
  if (need < 0 || bytes != need) return RG_ERR_ARG;
  const int G = e->G, P = e->P, L = e->L, PP = e->PP;
  HIPCHK(hipStreamSynchronize(e->stream));
  int rc;
  int shift; unsigned mask;
  if (field < RG_F_PEER_I32_END) {
    const bool isFlag = flag_bit_of(field, &shift, &mask) != 0;
    int *dev = isFlag ? e->S.flags : peer_word(e, field);
    if (!dev) return RG_ERR_ARG;
    std::vector<int> tmp((size_t)e->nLanes);
    rc = d2h(e, tmp.data(), dev, tmp.size() * 4);
    if (rc != RG_OK) return rc;
    int32_t *b = (int32_t *)buf;
    for (int g = 0; g < G; g++) for (int p = 0; p < P; p++) {
      int &wv = tmp[(size_t)g * PP + p];
      int32_t &hv = b[(size_t)g * P + p];
      if (!wr) hv = isFlag ? (int)(((unsigned)wv >> shift) & mask) : wv;
      else if (isFlag) wv = (int)(((unsigned)wv & ~(mask << shift)) | (((unsigned)hv & mask) << shift));
      else wv = hv;
    }
    if (wr) { rc = h2d(e, dev, tmp.data(), tmp.size() * 4); if (rc != RG_OK) return rc; e->derivedDirty = true; }
    return RG_OK;
  }
  if (field == RG_F_APPLYHASH) {
    std::vector<u64> tmp((size_t)e->nLanes);
    rc = d2h(e, tmp.data(), e->S.applyHash, tmp.size() * 8);
    if (rc != RG_OK) return rc;
    u64 *b = (u64 *)buf;
    for (int g = 0; g < G; g++) for (int p = 0; p < P; p++) { if (wr) tmp[(size_t)g * PP + p] = b[(size_t)g * P + p]; else b[(size_t)g * P + p] = tmp[(size_t)g * PP + p]; }
    if (wr) return h2d(e, e->S.applyHash, tmp.data(), tmp.size() * 8);
    return RG_OK;
  }
  if (field == RG_F_NEXTINDEX || field == RG_F_MATCHINDEX) {
    int *dev = field == RG_F_NEXTINDEX ? e->S.nextIdx : e->S.matchIdx;
    std::vector<int> tmp((size_t)e->Gpad * PP * PP);
    rc = d2h(e, tmp.data(), dev, tmp.size() * 4);
    if (rc != RG_OK) return rc;
    int32_t *b = (int32_t *)buf;
    for (int g = 0; g < G; g++) for (int p = 0; p < P; p++) for (int j = 0; j < P; j++) {
      int &wv = tmp[((size_t)g * PP + p) * PP + j]; int32_t &hv = b[((size_t)g * P + p) * P + j];
      if (wr) wv = hv; else hv = wv;
    }
    if (wr) { rc = h2d(e, dev, tmp.data(), tmp.size() * 4); if (rc != RG_OK) return rc; e->derivedDirty = true; }
    return RG_OK;
  }
  if (field == RG_F_LOGTERM || field == RG_F_SNAPLOGTERM || field == RG_F_LOGCMD || field == RG_F_SNAPLOGCMD) {
    const bool snap = field == RG_F_SNAPLOGTERM || field == RG_F_SNAPLOGCMD;
    const bool isCmd = field == RG_F_LOGCMD || field == RG_F_SNAPLOGCMD;
    std::vector<int> flags((size_t)e->nLanes);
    if (wr) {
      // a host write to either array first gives every persisted log its own rows, so that writing
      // the log does not change what is persisted and vice versa
      rc = ensure_snap_rows(e);
      if (rc != RG_OK) return rc;
      hipLaunchKernelGGL(k_materialize_snap, dim3((unsigned)((e->nLanes + 255) / 256)), dim3(256), 0, e->stream, e->S, e->dc, e->PP);
      HIPCHK(hipGetLastError());
      if (snap) {
        rc = d2h(e, flags.data(), e->S.flags, flags.size() * 4);
        if (rc != RG_OK) return rc;
        for (auto &f : flags) f |= (int)F_SNAPEXT;
        rc = h2d(e, e->S.flags, flags.data(), flags.size() * 4);
        if (rc != RG_OK) return rc;
      }
    } else if (snap) {
      rc = d2h(e, flags.data(), e->S.flags, flags.size() * 4);
      if (rc != RG_OK) return rc;
    }
    char *dev = (snap && wr) ? e->S.snapRows : e->S.logRows;
    const bool both = snap && !wr && e->S.snapRows != nullptr;   // reading persisted logs: per peer from where they live
    // group-sized chunks keep the staging buffers small
    const int chunk = std::max(1, (int)std::min<int64_t>(G, (64ll << 20) / ((int64_t)L * e->rowBytes)));
    std::vector<char> rows((size_t)chunk * L * e->rowBytes), rows2(both ? rows.size() : 0);
    for (int g0 = 0; g0 < G; g0 += chunk) {
      const int ng = std::min(chunk, G - g0);
      const size_t nb = (size_t)ng * L * e->rowBytes;
      rc = d2h(e, rows.data(), dev + (size_t)g0 * L * e->rowBytes, nb);
      if (rc != RG_OK) return rc;
      if (both) { rc = d2h(e, rows2.data(), e->S.snapRows + (size_t)g0 * L * e->rowBytes, nb); if (rc != RG_OK) return rc; }
      for (int g = 0; g < ng; g++) for (int p = 0; p < P; p++) {
        const bool ext = both && (((unsigned)flags[(size_t)(g0 + g) * PP + p]) & F_SNAPEXT);
        char *base = ext ? rows2.data() : rows.data();
        for (int i = 0; i < L; i++) {
          char *row = base + ((size_t)g * L + i) * e->rowBytes;
          const size_t hi = ((size_t)(g0 + g) * P + p) * L + i;
          if (isCmd) { u64 *hv = (u64 *)buf + hi; if (wr) memcpy(row + e->cmdOff + 8 * (size_t)p, hv, 8); else memcpy(hv, row + e->cmdOff + 8 * (size_t)p, 8); }
          else { int32_t *hv = (int32_t *)buf + hi; if (wr) memcpy(row + 4 * (size_t)p, hv, 4); else memcpy(hv, row + 4 * (size_t)p, 4); }
        }
      }
      if (wr) { rc = h2d(e, dev + (size_t)g0 * L * e->rowBytes, rows.data(), nb); if (rc != RG_OK) return rc; }
    }
    if (wr) { e->derivedDirty = true; e->syncDirty = true; }
    return RG_OK;
  }
  if (field >= RG_F_STARTSEQ && field <= RG_F_GFLAGS) {
    int *dev = group_word(e, field);
    if (wr) return h2d(e, dev, buf, (size_t)G * 4);
    return d2h(e, buf, dev, (size_t)G * 4);
  }
  if (field == RG_F_APPLYREC) {
    if (wr) return h2d(e, e->S.applyRec, buf, (size_t)G * L * 8);
    return d2h(e, buf, e->S.applyRec, (size_t)G * L * 8);
  }
  return RG_ERR_ARG;
}

extern "C" int rg_read_state(rg_engine *e, int32_t field, void *dst, int64_t bytes) { LOCK(e); return rw_state(e, field, dst, bytes, false); }
extern "C" int rg_write_state(rg_engine *e, int32_t field, const void *src, int64_t bytes) { LOCK(e); return rw_state(e, field, (void *)src, bytes, true); }

extern "C" int rg_state_hash(rg_engine *e, uint64_t *out, int64_t n) {
  LOCK(e);
  if (!out || n != e->G) return RG_ERR_ARG;
  TmpBuf tb;
  u64 *d = tb.get<u64>((size_t)e->G);
  if (!d) return RG_ERR_NOMEM;
  const int blocks = (int)((e->nLanes + 255) / 256);
  if (e->PP == 4) hipLaunchKernelGGL(k_hash<4>, dim3(blocks), dim3(256), 0, e->stream, e->S, e->dc, d);
  else hipLaunchKernelGGL(k_hash<8>, dim3(blocks), dim3(256), 0, e->stream, e->S, e->dc, d);
  HIPCHK(hipGetLastError());
  return d2h(e, out, d, (size_t)e->G * 8);
}

extern "C" int rg_counters(rg_engine *e, uint64_t *out) {
  LOCK(e);
  if (!out) return RG_ERR_ARG;
  int rc = d2h(e, out, e->S.counters, RG_N_COUNTERS * 8);
  if (rc != RG_OK) return rc;
  out[RG_C_GROUP_TICKS] = (uint64_t)e->G * (uint64_t)e->now;
  return RG_OK;
}

extern "C" int rg_rpc_bytes(rg_engine *e, int64_t command_bytes, uint64_t *total) {
  if (!e || !total || command_bytes < 0) return RG_ERR_ARG;
  if (!e->dc.acctBytes) return RG_ERR_STATE;
  uint64_t c[RG_N_COUNTERS];
  int rc = rg_counters(e, c);
  if (rc != RG_OK) return rc;
  *total = c[RG_C_RPC_FIXED_BYTES] + (4 + (uint64_t)command_bytes) * c[RG_C_AE_ENTRIES_SENT];
  return RG_OK;
}

extern "C" int rg_safety_check(rg_engine *e) {
  LOCK(e);
  const long long n = (long long)e->G * e->L;
  hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->S, e->dc, e->PP);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_safety_flags(rg_engine *e, uint32_t *out) {
  LOCK(e);
  if (!out) return RG_ERR_ARG;
  HIPCHK(hipMemsetAsync(e->S.flagSummary, 0, 8, e->stream));
  hipLaunchKernelGGL(k_flag_summary, dim3((e->G + 255) / 256), dim3(256), 0, e->stream, e->S, e->G);
  HIPCHK(hipGetLastError());
  return d2h(e, out, e->S.flagSummary, 8);
}

extern "C" int rg_device_buffers(rg_engine *e, void **counters_dev, void **flag_summary_dev) {
  LOCK(e);
  if (counters_dev) *counters_dev = e->S.counters;
  if (flag_summary_dev) *flag_summary_dev = e->S.flagSummary;
  return RG_OK;
}

#if defined(RG_SECTION_COUNTS) || defined(RG_STAMPS)
// profiling builds only: how often each section of k_tick ran, per wave
extern "C" int rg_debug_section_counts(rg_engine *e, uint64_t *out /* RG_CTR_SLOTS */) {
  LOCK(e);
  if (!out) return RG_ERR_ARG;
  return d2h(e, out, e->S.counters, RG_CTR_SLOTS * 8);
}
#endif
