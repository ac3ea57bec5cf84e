// rg_state.h -- device state of the engine and the reference's statements as per-lane functions
// (included by raftgpu.hip after <hip/hip_runtime.h>, include/raftgpu.h and the u64 typedef)
#ifndef RG_STATE_H
#define RG_STATE_H

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
  int *schedPos, *schedRegs;     // [Gpad] scenario schedule: next op, the group's peer registers
  // [Gpad][2] launches that only touch the groups that are due: word 0, wake = the group's next tick with something to do, as
  // its last span left it (0: unknown, look again); word 1, settled = the tick up to which the group's lazy work
  // (election-loop and monitor wake-up chains, client Starts) has been carried out.  One 8-byte pair per group, written by
  // one store at the end of a span: blocks of a due launch that share a window must agree on which of its groups were due
  // when the launch began, and a pair shows that either way -- wake below the launch's end (not run yet) or settled equal
  // to it (run by this very launch).
  int *wakeSettled;
#define RG_WAKE(S, g) (S).wakeSettled[2 * (size_t)(g)]
#define RG_SETTLED(S, g) (S).wakeSettled[2 * (size_t)(g) + 1]
  unsigned *gflags;              // [Gpad]
  u64 *counters;                 // [RG_N_COUNTERS]
  unsigned *flagSummary;         // [2]
};

struct DevCfg {
  int G, P, L, rowBytes, cmdOff;
  int cols, groupBytes;          // log layout: 0 = rows (entry i of all peers in one row), 1 = columns (per-peer arrays); bytes of one group's log
  long long groupBase;
  u64 seedKey;                   // sm64(seed)
  int reqDrop, replyDrop, replyDelay;
  int startPeriod, startFirst, startMax, startMode;
  int partPeriod, partJitter;
  int churnPeriod, churnDisc, churnRestart, churnCrash;
  int acctBytes;
  int nbA, gwA, gwB;             // tick kernel: sets [0, nbA) hold gwA groups per wave, the others gwB
  int relaySets, relayChunk;     // relay launch: number of sets (0: plain launch) and ticks per chunk
  int lazy;                      // some groups were skipped by earlier launches: a span starts from the group's `settled` tick
  int dueWin, dueSplit;          // due launch: groups per block window (<= 256); blocks that share a window's rounds
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
#define F_ZOMBIE    (1u << 15)   // with F_DEAD: killed by Kill() alone (raft.go:377-380), its server is still on the net

#define STREAM_ELECTION 1ull
#define STREAM_NET      2ull
#define STREAM_CMD      3ull
#define STREAM_PART     4ull
#define STREAM_CHURN    5ull
#define STREAM_SCHED    6ull

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
#define RG_CTR_SLOTS 56   // device counter block: RG_N_COUNTERS public counters, then profiling slots of the diagnostic builds
#if defined(RG_SECTION_COUNTS) || defined(RG_STAMPS) || defined(RG_DSTAMPS)
#define RG_WAVE_CTR_SLOTS RG_CTR_SLOTS
#else
#define RG_WAVE_CTR_SLOTS RG_N_COUNTERS   // a wave's LDS copy holds the public counters only
#endif
struct WaveCtr { unsigned c[RG_WAVE_CTR_SLOTS]; };
#ifdef RG_SECTION_COUNTS
#define SECTION_COUNT(w, i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&(w)->c[RG_N_COUNTERS + (i)], 1u); } while (0)
#define SECTION_COUNT_LANE(w, i) atomicAdd(&(w)->c[RG_N_COUNTERS + (i)], 1u)   /* once per calling lane */
#else
#define SECTION_COUNT(w, i) do { } while (0)
#define SECTION_COUNT_LANE(w, i) do { } while (0)
#endif
#ifdef RG_MARKS    // diagnostic builds only: comment lines in the assembly at the phase boundaries, for static instruction counts
#define STAMP(k) asm volatile("; MARK " #k)
#define STAMP_INIT asm volatile("; MARK begin")
#elif defined(RG_STAMPS)   // diagnostic builds only: shader cycles per phase of a tick, summed per wave into slots 40..
#define STAMP(k) do { unsigned long long _n = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) atomicAdd(&w->c[40 + (k)], (unsigned)(_n - _stamp)); _stamp = _n; } while (0)
#define STAMP_INIT unsigned long long _stamp = __builtin_readcyclecounter()
#else
#define STAMP(k) do { } while (0)
#define STAMP_INIT do { } while (0)
#endif
#ifdef RG_DSTAMPS          // diagnostic builds only: shader cycles of the stretches of one span (set-up, first look, the passes, wind-up), slots 40..
#define DSTAMP(k) do { unsigned long long _n = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) atomicAdd(&w->c[40 + (k)], (unsigned)(_n - _dstamp)); _dstamp = _n; } while (0)
#define DSTAMP_INIT unsigned long long _dstamp = __builtin_readcyclecounter()
#else
#define DSTAMP(k) do { } while (0)
#define DSTAMP_INIT do { } while (0)
#endif

enum { CD_SNAPTERM = 0, CD_SNAPVOTE, CD_SNAPLEN, CD_RNGCTR, CD_ROUNDSEQ, CD_AH_LO, CD_AH_HI, CD_LASTTERM, CD_FIRSTOWN, CD_CTERM, CD_N };

typedef __attribute__((address_space(3))) int lds_int;

// Registers of one peer.  CP = type of the pointer to the event-only words (LDS rows or a local array).
template <typename CP>
struct LaneT {
  int term, vote, commit, logLen, timer, timeout, hbTimer, elNext, monNext;
  int applied;                   // lastApplied: looked at in every pass (is there anything to apply?)
  int snapSync;                  // entries of the log the persisted copy already holds: looked at by every persist()
  unsigned flags;
  // Words touched only by events (persist, draws, round numbering, the apply digest, and three caches of
  // log words rebuilt by k_derive after host writes: term of the last entry, first index from which every entry
  // carries the current term, term of the entry at commitIndex) live outside the register
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
  // can a Call reach this peer's handler: alive, or only Kill()ed (the handler then returns at once, raft.go:203,273)
  __device__ bool served() const { return (flags & (F_DEAD | F_ZOMBIE)) != F_DEAD; }
  // crash1, config.go:99-127: off the net, server deleted, dead
  __device__ void crash() { flags = (flags & ~(F_CONNECTED | F_ZOMBIE)) | F_DEAD; }
  // Kill(), raft.go:377-380: the dead flag alone
  __device__ void kill() { if (!(flags & F_DEAD)) flags |= F_DEAD | F_ZOMBIE; }

  __device__ void load(const DevPtrs &S, long long gl) {
    term = S.term[gl]; vote = S.vote[gl]; flags = (unsigned)S.flags[gl]; commit = S.commit[gl]; applied = S.applied[gl];
    logLen = S.logLen[gl]; timer = S.timer[gl]; timeout = S.timeout[gl]; hbTimer = S.hbTimer[gl]; elNext = S.elNext[gl];
    monNext = S.monNext[gl]; cd(CD_RNGCTR) = S.rngCtr[gl]; cd(CD_ROUNDSEQ) = S.roundSeq[gl];
    cd(CD_SNAPTERM) = S.snapTerm[gl]; cd(CD_SNAPVOTE) = S.snapVote[gl]; cd(CD_SNAPLEN) = S.snapLen[gl]; snapSync = S.snapSync[gl];
    cd(CD_LASTTERM) = S.lastTerm[gl]; cd(CD_FIRSTOWN) = S.firstOwn[gl]; cd(CD_CTERM) = S.commitTerm[gl]; setApplyHash(S.applyHash[gl]);
  }
  __device__ void store(const DevPtrs &S, long long gl) const {
    S.term[gl] = term; S.vote[gl] = vote; S.flags[gl] = (int)flags; S.commit[gl] = commit; S.applied[gl] = applied;
    S.logLen[gl] = logLen; S.timer[gl] = timer; S.timeout[gl] = timeout; S.hbTimer[gl] = hbTimer; S.elNext[gl] = elNext;
    S.monNext[gl] = monNext; S.rngCtr[gl] = cd(CD_RNGCTR); S.roundSeq[gl] = cd(CD_ROUNDSEQ);
    S.snapTerm[gl] = cd(CD_SNAPTERM); S.snapVote[gl] = cd(CD_SNAPVOTE); S.snapLen[gl] = cd(CD_SNAPLEN); S.snapSync[gl] = snapSync;
    S.lastTerm[gl] = cd(CD_LASTTERM); S.firstOwn[gl] = cd(CD_FIRSTOWN); S.commitTerm[gl] = cd(CD_CTERM); S.applyHash[gl] = applyHash();
  }
};

// Addressing of one group's log.  Two layouts, fixed per engine:
//   rows    -- entry i of ALL peers of the group is one row {int32 term[P]; uint64 cmd[P]} of rowBytes bytes, rows
//              consecutive in i: replicating entry i to the followers and applying index i touch one line;
//   columns -- per peer, int32 term[L] then (at cmdOff) uint64 cmd[L], contiguous in i: a replay of one peer's log
//              (raft.go:672-688 after a restart), a catch-up copy and the conflict walks (raft.go:308-313, 495-502) read
//              64 consecutive entries per request instead of one word per row.  (Long logs, crash / restart churn.)
// base is wave-uniform in the tick kernel (the block's first group), off the lane's 32-bit byte offset from it, so
// accesses use scalar-base + vector-offset addressing.  `cols` is a compile-time constant there (template parameter).
struct Rows {
  char *base; unsigned off; int rowBytes, cmdOff; int L; bool cols;
  // byte offset of a term / command word from the group's first byte
  __device__ unsigned toff(int i, int col) const {
    return cols ? ((unsigned)col * (unsigned)L + (unsigned)i) * 4u : (unsigned)i * (unsigned)rowBytes + (unsigned)col * 4u;
  }
  __device__ unsigned coff(int i, int col) const {
    return (unsigned)cmdOff + (cols ? ((unsigned)col * (unsigned)L + (unsigned)i) * 8u : (unsigned)i * (unsigned)rowBytes + (unsigned)col * 8u);
  }
  __device__ int *termp(int i, int col) const { return (int *)(base + (size_t)(off + toff(i, col))); }
  __device__ u64 *cmdp(int i, int col) const { return (u64 *)(base + (size_t)(off + coff(i, col))); }
  // the same for another lane's group (offB: that lane's `off`), for loops in which the whole wave serves one lane
  __device__ int *termp_at(unsigned offB, int i, unsigned col) const { return (int *)(base + (size_t)(offB + toff(i, (int)col))); }
  __device__ u64 *cmdp_at(unsigned offB, int i, unsigned col) const { return (u64 *)(base + (size_t)(offB + coff(i, (int)col))); }
};
__host__ __device__ static inline size_t group_log_bytes(const DevCfg &c) { return c.cols ? (size_t)c.groupBytes : (size_t)c.L * c.rowBytes; }
__device__ static inline Rows rows_of(char *arr, const DevCfg &c, int g) {
  Rows r; r.base = arr + (size_t)g * group_log_bytes(c); r.off = 0; r.rowBytes = c.rowBytes; r.cmdOff = c.cmdOff; r.L = c.L; r.cols = c.cols != 0; return r;
}
// the arrays the tick loop itself touches; everything else is read once and written once
struct HotPtrs {
  char *logRows, *snapRows; int *nextIdx, *matchIdx; u64 *applyRec;
  int *relay;                    // relay launches: [0] next item, [1 + set] chunks of the set that are done
  int *relayStage;               // relay launches: per set, 7 words per group (the 5 per-group words, rounds sent before and in the last span), padded to whole 128-B lines
  rg_apply_msg *applyRing;       // [G * P][applyCap]: the ApplyMsgs each peer emitted, in order, slot = message number % applyCap
  unsigned *applyHead;           // [Gpad * PP], indexed like the per-peer words: messages emitted so far
  int applyCap;                  // 0: no ring (nobody reads applyCh from the host)
  const rg_sched_op *sched; int schedN;   // the scenario schedule (sorted by tick), or null / 0
  const unsigned *netCfg;        // [Gpad] per-group net rates (bit 31: set; 10 bits each: request drop, reply drop, reply delay), or null
};

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

// first tick whose time is >= ms (ms >= 0)
#define MS_TICK(ms) ((int)(((unsigned)((ms) > 0 ? (ms) : 0) + (unsigned)(RG_TICK_MS - 1)) / (unsigned)RG_TICK_MS))
// scheduler slack of the election loop's 50 ms sleep: 0..3 ms, seeded
__device__ static inline int sleep_slack(u64 key, int p, int now) {
  unsigned h = (unsigned)key ^ ((unsigned)p * 0x9E3779B1u) ^ ((unsigned)now * 0x85EBCA6Bu);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return (int)(h >> 30);
}
// Wake-ups of a peer that lie before tick T and change nothing but their own next wake-up time (raft.go:548-563 and 664-667
// for a follower whose timer has not run out and for a leader; raft.go:421, 524 for a leader's poll behind a closed gate):
// the tick kernel does not run them one by one but advances the two chains when the peer's group next has something to do,
// and so does a host verb that changes whether or how the chains go on (crash, Kill) before it takes effect.
template <class LN>
__device__ static inline void catch_up_lane(LN &r, u64 key, int peer, int T) {
  if (r.has(F_DEAD)) return;
  const int st = r.state();
  if (st != RG_CANDIDATE) {                                   // a candidate's wake-ups are rounds: never skipped
    const int extra = st == RG_LEADER ? RG_ELECTION_LEADER_EXTRA_MS : 0;
    int tau = MS_TICK(r.elNext);
    while (tau < T) {
      const int ms = tau * RG_TICK_MS;
      r.elNext = ms + RG_ELECTION_PERIOD_MS + extra + sleep_slack(key, peer, ms);
      tau = MS_TICK(r.elNext);
    }
  }
  if (r.has(F_MONALIVE) && r.isLeader()) {                    // polls behind a closed gate: every second tick
    const int tau = MS_TICK(r.monNext);
    if (tau < T) r.monNext = (tau + ((T - 1 - tau) & ~1)) * RG_TICK_MS + RG_MONITOR_POLL_MS;
  }
}

// Make() + readPersist() + the head of election(), raft.go:700-731,129-152,538-547.
// The harness's start1 runs crash1 first (config.go:137), so the peer comes
// back disconnected.
template <class LN>
__device__ static inline void make_peer(LN &r, const Rows &lg, const Rows &sn, int col, int L, u64 key, int now, WaveCtr *w) {
  r.flags &= ~(F_DEAD | F_ZOMBIE | F_STATE_MASK | F_STOP | F_CNT_MASK | F_MONALIVE | F_FIRST | F_RESTART | F_CDIRTY);
  r.term = 0; r.vote = -1;
  *lg.termp(0, col) = 0; *lg.cmdp(0, col) = 0;
  r.logLen = 1; r.commit = 0; r.applied = 0; r.timer = now; r.hbTimer = now;
  r.cd(CD_LASTTERM) = 0; r.cd(CD_CTERM) = 0;
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
    r.cd(CD_LASTTERM) = *lg.termp(r.logLen - 1, col);
    WAIT_LOADS();
    r.cd(CD_CTERM) = r.cd(CD_LASTTERM);
  }
  r.monNext = 0;
  r.timeout = 0;
  r.cd(CD_FIRSTOWN) = r.logLen;
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
  r.cd(CD_LASTTERM) = r.term;
  if (idx == r.commit) r.cd(CD_CTERM) = r.term;       // only when commitIndex already ran past the log
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
// SCAN = false leaves the walk back to the first index of the conflicting term (raft.go:308-313) to the caller:
// ci is then RG_CI_PENDING.
#define RG_CI_PENDING (-0x7fffffff)
template <bool SCAN = true, class LN>
__device__ static inline AppendDecision decide_append(const LN &r, const Rows &lg, int col, int aTerm, int prev, int prevTerm, int myPrevTerm) {
  AppendDecision d; d.ct = 0; d.ci = 0; d.match = false;
  if (aTerm < r.term) { d.rTerm = r.term; d.touch = false; return d; }
  d.touch = true;
  d.rTerm = aTerm > r.term ? aTerm : r.term;
  if (prev + 1 <= r.logLen && myPrevTerm == prevTerm) { d.match = true; return d; }
  if (prev + 1 > r.logLen) { d.ct = -1; d.ci = r.logLen; return d; }
  d.ct = myPrevTerm;
  if constexpr (!SCAN) { d.ci = RG_CI_PENDING; return d; }
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

#endif
