// rg_tick.h -- the tick kernel: n lockstep ticks of every group in one launch (needs rg_state.h)
#ifndef RG_TICK_H
#define RG_TICK_H

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

// Dense variant for the tick kernel: a group's P peers sit in P adjacent lanes, up to 64/P whole groups per
// wave (12 for P = 5), instead of padding every group to a power of two.  A wave may be given fewer groups
// (gw): the cost of a tick grows with the number of groups that share a wave, so when the groups of a launch
// (or of its last, partial batch) do not fill the chip's wave slots they are spread over more, thinner waves.
template <int P>
struct GrpN {
  static constexpr int GWMAX = 64 / P;       // most groups a wave can hold
  int lane, base, peer;
  bool inWave;
  __device__ explicit GrpN(int gw) {
    lane = (int)(threadIdx.x & 63);
    const int q = lane / P;
    base = q * P;
    peer = lane - base;
    inWave = q < gw;
  }
  __device__ int rot(int d) const { int x = peer + d; return base + (x >= P ? x - P : x); }
  __device__ int shfl(int v, int j) const { return __builtin_amdgcn_ds_bpermute((base + j) << 2, v); }
  __device__ unsigned shflu(unsigned v, int j) const { return (unsigned)__builtin_amdgcn_ds_bpermute((base + j) << 2, (int)v); }
  __device__ u64 shfl64(u64 v, int j) const {
    unsigned lo = shflu((unsigned)v, j), hi = shflu((unsigned)(v >> 32), j);
    return ((u64)hi << 32) | lo;
  }
  __device__ unsigned ballot(bool p) const { return (unsigned)(__builtin_amdgcn_ballot_w64(p) >> base) & ((1u << P) - 1u); }
  // Reductions over the group: every lane reads every other peer's value, P - 1 exchanges that do not wait for one another --
  // one trip through the LDS pipe in the lane's chain of dependent instructions, where doubling rotations make ceil(log2 P) (five
  // peers: 39.6 -> 39.0 ms; seven: 54.6 -> 54.3 ms).
  __device__ int gmin(int v) const {
    int t = v;
#pragma unroll
    for (int d = 1; d < P; d++) { const int o = __builtin_amdgcn_ds_bpermute(rot(d) << 2, v); t = o < t ? o : t; }
    return t;
  }
  __device__ int gmax(int v) const {
    int t = v;
#pragma unroll
    for (int d = 1; d < P; d++) { const int o = __builtin_amdgcn_ds_bpermute(rot(d) << 2, v); t = o > t ? o : t; }
    return t;
  }
  __device__ unsigned gor(unsigned v) const {
    unsigned t = v;
#pragma unroll
    for (int d = 1; d < P; d++) t |= (unsigned)__builtin_amdgcn_ds_bpermute(rot(d) << 2, (int)v);
    return t;
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
// network outcome of one Call, labrpc.go:216-307: 0 lost, 1 handler ran / reply lost or late, 2 round trip;
// bit 2 (NET_WIRE): the net carried the reply, in time or not, and counted its bytes (labrpc.go:286-290).
// cost: how long the sender's select (raft.go:469-474, 605-609) waits for this Call: until Call returns -- with a
// reply, or false right after the net's short delay when the net drops the request or the reply (labrpc.go:226-229,
// 273-275) -- or 25 ms, whichever is first.  One hash per Call holds its three draws and its delay, so every target
// works out its own share of the round's duration and the sender adds them up.
#define NET_WIRE 4
__device__ static inline int permille_of(unsigned field16) { return (int)((field16 * 1000u) >> 16); }
struct NetRates { int reqDrop, replyDrop, replyDelay; };     // the net model's three rates, permille
// per-group override word: bit 31 set, then 10 bits each
__device__ static inline NetRates unpack_rates(unsigned w) { NetRates n; n.reqDrop = (int)(w & 1023u); n.replyDrop = (int)((w >> 10) & 1023u); n.replyDelay = (int)((w >> 20) & 1023u); return n; }
__device__ static inline int net_outcome(const NetRates &c, u64 key, bool reachable, int s, int j, unsigned seq, int &cost) {
  cost = RG_RPC_WAIT_MS;
  const bool quiet = (c.reqDrop | c.replyDrop | c.replyDelay) == 0;
  if (reachable && quiet) { cost = 0; return 2 | NET_WIRE; }
  const u64 h = stream_hash(key, STREAM_NET, ((unsigned)s << 8) | (unsigned)j, seq);
  const unsigned lo = (unsigned)h, hi = (unsigned)(h >> 32);
  if (!reachable) {                               // labrpc.go:290-305: Call returns false after 0..6999 ms
    const int ms = (int)(((u64)hi * 7000ull) >> 32);
    cost = ms < RG_RPC_WAIT_MS ? ms : RG_RPC_WAIT_MS;
    return 0;
  }
  const int d = (int)((((hi >> 16) & 0xFFu) * (unsigned)RG_NET_DELAY_SPAN_MS) >> 8);          // labrpc.go:222, 0..26
  const int back = d < RG_RPC_WAIT_MS ? d : RG_RPC_WAIT_MS;
  if (permille_of(lo & 0xFFFFu) < c.reqDrop) { cost = back; return 0; }
  if (permille_of(lo >> 16) < c.replyDrop) { cost = back; return 1; }
  if (d >= RG_RPC_WAIT_MS) return 1 | NET_WIRE;   // the handler runs, the sender has given up
  if (permille_of(hi & 0xFFFFu) < c.replyDelay) return 1 | NET_WIRE;
  cost = d;
  return 2 | NET_WIRE;
}

// first peer of a tick's actor order
__device__ static inline int turn_start(u64 key, int tick, int P) {
  unsigned h = (unsigned)tick * 0x9E3779B1u + (unsigned)key;
  h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
  return (int)(((h >> 16) * (unsigned)P) >> 16);
}

#ifndef RG_VERIFY_BATCH
#define RG_VERIFY_BATCH 48   // applied entries a peer may have unchecked before the wave checks them
#endif
#ifndef RG_TICK_WAVES
#define RG_TICK_WAVES 4   // waves per SIMD the register allocation must allow
#endif
// Ticks [t0, t0 + nticks) of one block's worth of groups (`set`: what blockIdx.x is in a plain launch), by the
// whole block.  EXT: the extended instantiation, with wire-byte accounting (rg_config.account_bytes) and the
// apply ring (rg_config.apply_ring_cap) compiled in; the plain instantiation carries none of it.  LEAN: the
// instantiation for engines without churn and with the first-leader Start mode (the benchmark's), with that code left out
// (fewer registers to keep alive).  w: the calling wave's counters of the launch.  stIn / stOut: the set's group words as the
// previous span left them / for the next span, when spans of one launch run on different blocks (else null).
//
// DUE: the span of a due launch (k_tick_due).  `set` is then the first group of the calling WAVE's window, dueSlot the
// lane's group as an offset from it and dueValid whether the lane has a group at all; waves of a block work through
// different numbers of such spans, so nothing in here may wait for the block.  dueWake: the group's wake word as the block
// read it when it looked for its due groups (nobody else runs the group in this launch).
template <int P, bool EXT, bool LEAN, bool COLS, bool DUE = false>
__device__ __forceinline__ void tick_span(const DevPtrs *Sp, const HotPtrs &H, const DevCfg &c, const int set, const int t0, const int nticks, WaveCtr *w,
                                          const int *stIn, int *stOut, const int dueSlot = 0, const bool dueValid = false, const int dueWake = 0) {
  constexpr int PP = P <= 4 ? 4 : 8;        // stride of a group in the per-peer arrays and matrices in memory
  constexpr int NONE = 8;                   // "no such peer" for first-set-bit results
  constexpr int GPB = 4 * GrpN<P>::GWMAX;   // most groups of a 256-thread block
  // Runs of log entries are handled by the whole wave for one lane at a time (one memory round trip per lane, whatever
  // the run's length).  In a launch of a few ticks the LDS rings are cold and nearly every run is a single entry: there
  // every lane handles its own short run (up to SHORTN entries), so the wave pays one round trip for all of them.
#ifndef RG_SHORT_PATHS
#define RG_SHORT_PATHS DUE
#endif
  constexpr bool SHORTP = RG_SHORT_PATHS;
#ifndef RG_SHORTN
#define RG_SHORTN 4
#endif
  constexpr int SHORTN = RG_SHORTN;
  // sets [0, nbA) hold gwA groups per wave, the rest gwB
  const bool classA = DUE || set < c.nbA;
  const int gw = DUE ? GrpN<P>::GWMAX : (classA ? c.gwA : c.gwB);
  const GrpN<P> G(gw);
  const int wave = (int)(threadIdx.x >> 6);
  DSTAMP_INIT;
#ifdef RG_DSTAMPS
  const unsigned long long _dstamp0 = _dstamp, _dreal0 = __builtin_amdgcn_s_memrealtime();
#endif

  const int g0 = DUE ? set : (classA ? set * 4 * c.gwA : c.nbA * 4 * c.gwA + (set - c.nbA) * 4 * c.gwB);   // the set's first group: always < G
#ifndef RG_RING
#define RG_RING 4
#endif
  constexpr int RING = RG_RING;
  __shared__ int ringIdx[RING][256];
  constexpr int GPBPAD = (GPB + 31) & ~31;
  // Which of the set's groups the lane works on.  A wave makes as many passes as its busiest group has events, and
  // how busy a group is changes slowly: from the second span of a launch on, the set's groups are dealt to its
  // waves in the order of the rounds they sent in the span before (stIn word 6), busiest first, so that busy
  // groups share waves and the other waves finish early.  (Groups are independent: any dealing gives the same result.)
  int slot = DUE ? dueSlot : wave * gw + (G.lane / P);
  if (!DUE && stIn) {                                // the same for the whole block
    int *const rate = &ringIdx[0][0], *const order = &ringIdx[RING / 2][0];   // scratch: the rings are not in use yet
    static_assert(GPB <= 256 * RING / 2, "scratch of the dealing");
    const int i = (int)threadIdx.x;
    if (i < GPB) rate[i] = (i < 4 * gw && g0 + i < c.G) ? stIn[6 * GPBPAD + i] : (int)0x80000000;   // groups past the set's or the last one: dealt last
    __syncthreads();
    if (i < GPB) {
      const int mine = rate[i];
      int rank = 0;
      for (int j = 0; j < GPB; j++) { const int o = rate[j]; rank += (o > mine || (o == mine && j < i)) ? 1 : 0; }
      order[rank] = i;
    }
    __syncthreads();
    slot = order[slot < GPB ? slot : 0];
    __syncthreads();                         // the rings are put to their own use below
  }
  const int g = g0 + slot;
  const int peer = G.peer;
  const bool valid = DUE ? dueValid : (G.inWave && g < c.G);
  const int L = c.L;
  const unsigned gLoc = valid ? (unsigned)(g - g0) : 0u;       // in-range group for address arithmetic of idle lanes
  // the group's slot in the block's per-group LDS words
  const unsigned gSlot = DUE ? (unsigned)(wave * GrpN<P>::GWMAX + (G.inWave ? G.lane / P : 0)) : gLoc;
  const int gq = g0 + (int)gLoc;
  const long long gl = (long long)(G.inWave ? g : g0) * PP + (G.inWave ? peer : 0);   // slot in the per-peer arrays
  Rows lg, sn;
  const unsigned groupBytes = COLS ? (unsigned)c.groupBytes : (unsigned)L * (unsigned)c.rowBytes;
  lg.base = H.logRows + (size_t)g0 * groupBytes; lg.off = gLoc * groupBytes; lg.rowBytes = c.rowBytes; lg.cmdOff = c.cmdOff; lg.L = L; lg.cols = COLS;
  sn = lg; sn.base = H.snapRows + (size_t)g0 * groupBytes;
  int *const nextBlk = H.nextIdx + (size_t)g0 * PP * PP;       // uniform; rows [group][owner][j]
  int *const matchBlk = H.matchIdx + (size_t)g0 * PP * PP;
  const unsigned mOff = gLoc * (unsigned)(PP * PP);
#define nextRow(i) nextBlk[mOff + (unsigned)(i)]
#define matchRow(i) matchBlk[mOff + (unsigned)(i)]
  u64 *const recBlk = H.applyRec + (size_t)g0 * L;
  const unsigned gLocL = gLoc;
  const int peerL = peer;
#define rOff (gLocL * (unsigned)L)
#define rec(i) recBlk[rOff + (unsigned)(i)]
  u64 key = sm64(c.seedKey ^ (u64)(c.groupBase + g));
  // the net's rates: the engine's, or (extended instantiation) the group's own (Network.Reliable / LongReordering are
  // per tester config in the reference, config.go:288-294, labrpc.go:169-181)
  NetRates net; net.reqDrop = c.reqDrop; net.replyDrop = c.replyDrop; net.replyDelay = c.replyDelay;
  if constexpr (EXT) {
    if (H.netCfg) { const unsigned nw = H.netCfg[gq]; if (nw >> 31) net = unpack_rates(nw); }
  }

  __shared__ int coldLds[CD_N + (EXT ? 4 : 0)][256];   // the peers' event-only words; extended instantiation: per-lane byte sums, the schedule's position and registers
  __shared__ int churnNext[GPB];                              // per group: its next churn tick (written by all its lanes alike)
  // The last RING entries each peer wrote to its log, staged in LDS: a follower copies a leader's fresh entries,
  // and every peer applies its fresh entries, without a trip to HBM.  Slot = index % RING, tagged with the index; HBM is written through and stays authoritative.
  __shared__ int ringTerm[RING][256];
  __shared__ u64 ringCmd[RING][256];
  const int tid = (int)threadIdx.x;
#pragma unroll
  for (int q = 0; q < RING; q++) ringIdx[q][tid] = -1;
  if constexpr (!DUE) __syncthreads();     // (lanes only ever read ring columns of their own wave)
#define RING_PUSH(i_, tm_, cm_) do { const int _q = (i_) & (RING - 1); ringIdx[_q][tid] = (i_); ringTerm[_q][tid] = (tm_); ringCmd[_q][tid] = (cm_); } while (0)
  LaneT<lds_int *> r;
  r.cold = (lds_int *)&coldLds[0][threadIdx.x]; r.cs = 256;
  r.load(*Sp, gl);                       // tail lanes read a valid slot and never write it back
  // The group's words, a copy in each of its lanes, the apply cursor, the cursor of the checked applies and the lane's cached
  // matrix pair are looked at in every pass or every round: they stay in registers (as LDS rows, each look was a trip through
  // the LDS pipe in the middle of a chain of dependent instructions: 40.9 against 39.7 ms).
  int gStartSeq, gPartEpoch, gNextStart, gPartNext, gRecMax;
  gStartSeq = stIn ? stIn[gLoc] : Sp->startSeq[gq]; gPartEpoch = stIn ? stIn[GPBPAD + gLoc] : Sp->partEpoch[gq];
  int lVerified = r.applied;
#ifdef RG_DSTAMPS
  WAIT_LOADS();
#endif
  DSTAMP(1);
  if (stOut) {                               // rounds the group's peers have sent so far (see the end of the span)
    const int sent = G.gsum(valid ? r.cd(CD_ROUNDSEQ) : 0);
    if (valid && peer == 0) stOut[5 * GPBPAD + gLoc] = sent;
  }
  if constexpr (EXT) { coldLds[CD_N][threadIdx.x] = 0; coldLds[CD_N + 1][threadIdx.x] = 0; }
#define gSchedPos coldLds[CD_N + 2][threadIdx.x]
#define gRegs coldLds[CD_N + 3][threadIdx.x]
  if constexpr (EXT) { gSchedPos = H.schedN > 0 ? Sp->schedPos[gq] : 0; gRegs = H.schedN > 0 ? Sp->schedRegs[gq] : 0xFFFF; }
  gPartNext = stIn ? stIn[2 * GPBPAD + gLoc] : Sp->partNext[gq]; gRecMax = stIn ? stIn[3 * GPBPAD + gLoc] : Sp->recMax[gq];
  unsigned gfl = 0;
  // next ticks of the periodic control events (kept as counters: no division in the loop)
  // Within a launch each lane caches the nextIndex / matchIndex words ONE leader (cOwner) keeps for it; memory is
  // written through, so the matrices in HBM stay complete for other owners, dumps and digests.
  int cOwner = -1, cNext = 0, cMatch = 0;
  // stores to words another lane may read later are drained (vmcnt) only right before such a read
  bool storesPending = false;
#define SYNC_STORES() do { if (storesPending) { SECTION_COUNT(w, 7); wave_fence(); storesPending = false; } } while (0)
  // A span of a due launch meets empty rings and no cached matrix words, and its one or two passes would fetch them from
  // HBM one after the other in the middle of a round (the leader's words for this lane, the term at PrevLogIndex, the
  // entries).  Here they are asked for up front instead, all in one trip that the rest of the set-up overlaps: the words
  // the group's (first) leader keeps for this lane -- only the lane itself ever writes them -- and the last RING entries
  // of the lane's own log, which go into its ring right before the first pass.
  int pfTerm[RING]; u64 pfCmd[RING];
  bool pfLive = false;
  if constexpr (DUE) {
    pfLive = valid && !r.has(F_DEAD);
    const unsigned lm = G.ballot(pfLive && r.isLeader());
    const int s0 = lm ? (__ffs((int)lm) - 1) : 0;
    if (pfLive && lm != 0 && peer != s0) { cNext = nextRow(s0 * PP + peer); cMatch = matchRow(s0 * PP + peer); cOwner = s0; }
#pragma unroll
    for (int q = 0; q < RING; q++) {
      const int i = r.logLen - 1 - q;
      pfTerm[q] = 0; pfCmd[q] = 0;
      if (pfLive && i >= 0) { pfTerm[q] = *lg.termp(i, peer); pfCmd[q] = *lg.cmdp(i, peer); }
    }
  }
  // Groups never exchange anything, so a wave need not keep its groups at the same tick.  Every group has its own
  // clock tg: the next tick at which one of its peers or its control schedule has something to do.  One pass of
  // the loop below advances every unfinished group of the wave through ITS next such tick; ticks in which a
  // group has nothing to do are skipped for that group alone.  Wake-ups that provably change nothing but their
  // own next wake-up time (a follower whose timer has not run out, a leader's election loop, a leader's monitor
  // poll behind a closed gate) are not events: their chains are caught up when the group next runs a tick
  // (catch_up) and once more when the launch ends, so every word is as the tick-by-tick order leaves it.
  const int tEnd = t0 + nticks;
#define gChurnNext churnNext[gSlot]
  {
    int ns = 0x7fffffff;
    if (c.startPeriod > 0) {
      // client Starts before tick tb have been carried out: tb is this span's first tick, or -- after launches that left
      // the group alone -- the tick up to which its last span brought it
      const int tb = (c.lazy && !stIn) ? RG_SETTLED(*Sp, gq) : t0;
      ns = c.startFirst;
      if (tb > c.startFirst) ns += (tb - c.startFirst + c.startPeriod - 1) / c.startPeriod * c.startPeriod;
    }
    gNextStart = ns;
    // the group's next churn tick: the first positive multiple of the period in this span or after it (never skipped)
    // (only the group's own lanes write its word: idle lanes carry gLoc 0, another wave's group)
    if (valid) gChurnNext = (!LEAN && c.churnPeriod > 0) ? ((t0 > 1 ? t0 : 1) + c.churnPeriod - 1) / c.churnPeriod * c.churnPeriod : 0x7fffffff;
  }
  // wake-ups of this lane that lie before tick T and change nothing else
  auto catch_up = [&](int T) { if (valid) catch_up_lane(r, key, peerL, T); };
  // first tick >= from at which the lane's group has something to do
  auto next_event = [&](int from) -> int {
    int cand = 0x7fffffff;       // this lane's next event
    int candS = 0x7fffffff;      // the same if the pending client Start has been given to this lane first
    const bool startOn = c.startPeriod > 0 && gStartSeq < c.startMax;
    const int sTick = gNextStart;
    const bool alive = valid && !r.has(F_DEAD);
    // who takes the pending Start: every leader with room (test_test.go:697-703) or the first connected one (config.go:436-451)
    bool taker = startOn && alive && r.isLeader() && r.logLen < L && ((!LEAN && c.startMode == RG_START_ALL) || r.has(F_CONNECTED));
    if (c.startPeriod > 0 && !(!LEAN && c.startMode == RG_START_ALL)) {
      const unsigned tm = G.ballot(taker);
      taker = taker && peer == __ffs((int)tm) - 1;
    }
    if (alive) {
      const int st = r.state();
      const int e = MS_TICK(r.elNext);
      if (st == RG_CANDIDATE) cand = e;
      else if (st == RG_FOLLOWER) {
        const int fire = (r.timer + r.timeout) / RG_TICK_MS + 1;  // first tick with now - timer > timeout (raft.go:557)
        cand = e > fire ? e : fire;                               // if the chain is behind, tick `fire` only catches it up
      }
      candS = cand;
      if (r.has(F_MONALIVE)) {
        const int m = MS_TICK(r.monNext);
        int mGate = m, mStart = m;
        if (r.isLeader() && !(r.flags & (F_FIRST | F_CDIRTY))) {  // gate of raft.go:421; polls come every second tick
          const int open = MS_TICK(r.hbTimer + ((r.commit < r.logLen - 1) ? RG_HB_BACKLOG_MS : RG_HB_IDLE_MS));
          if (open > m) mGate = m + (((open - m + 1) >> 1) << 1);
          // with the Start's entry in the log from tick sTick on, the gate is the backlog one
          int openS = MS_TICK(r.hbTimer + RG_HB_BACKLOG_MS);
          openS = openS > sTick ? openS : sTick;
          if (openS > m) mStart = m + (((openS - m + 1) >> 1) << 1);
          if (!taker || mGate < mStart) mStart = mGate;
        }
        cand = mGate < cand ? mGate : cand;
        candS = mStart < candS ? mStart : candS;
      }
      if (r.commit > r.applied) { cand = from; candS = from; }   // entries committed by a host call are applied at once
      // The first connected leader takes the Starts only while its log has room: the Start that no longer fits goes to
      // the next leader (or to nobody), whose gate it changes, so the group runs at that Start's tick and looks again.
      if (taker && !(!LEAN && c.startMode == RG_START_ALL) && gStartSeq + (L - r.logLen) < c.startMax) {
        const int fullAt = sTick + (L - r.logLen) * c.startPeriod;
        candS = fullAt < candS ? fullAt : candS;
      }
    }
    cand = G.gmin(cand);
    if (c.partPeriod > 0 && gPartNext < cand) cand = gPartNext;
    if ((!LEAN && c.churnPeriod > 0) && gChurnNext < cand) cand = gChurnNext;
    if constexpr (EXT) {
      if (H.schedN > 0 && valid && gSchedPos < H.schedN) { const int st = H.sched[gSchedPos].tick; cand = st < cand ? st : cand; }
    }
    if (c.startPeriod > 0 && wave_any(startOn && sTick < cand)) {
      // The Start comes before anything else.  It changes no clock but its taker's gate (which then stays the
      // backlog one however many Starts follow), so it is not an event of its own: the group next runs when the
      // taker's (earlier) poll passes the gate.
      candS = G.gmin(candS);
      if (startOn && sTick < cand) cand = candS < cand ? candS : cand;
    }
    return cand > from ? cand : from;
  };

  // The tester's side of apply (config.go:170-194: compare the delivered command with the record of its index,
  // fold it into the peer's digest) looks at nothing but the entry and the record, and neither changes once
  // written unless the peer's log is rewritten below its apply cursor.  So it is not done entry by entry as the
  // cursor moves: lVerified trails r.applied, and the stretch between them is checked by the whole wave (one entry
  // per lane) before anything could overwrite one of its entries, and when the launch ends.
  auto verify_applied = [&](bool want) {
    const bool need = want && valid && lVerified < r.applied;
    u64 vm = __builtin_amdgcn_ballot_w64(need);
    if (vm) {
      SECTION_COUNT(w, 12);
      SYNC_STORES();
      const int nOwn = r.applied - lVerified;
      if constexpr (SHORTP) {
        const bool mine = need && nOwn <= SHORTN;
        if (mine) {
          u64 cmd[SHORTN], seen[SHORTN];
#pragma unroll
          for (int j = 0; j < SHORTN; j++) {           // all loads first (past the run's end: its last entry again)
            const int i = lVerified + 1 + (j < nOwn ? j : nOwn - 1), ic = i < L ? i : L - 1;
            cmd[j] = *lg.cmdp(ic, peer); seen[j] = rec(ic);
          }
          bool differs = false;
          unsigned accLo = 0, accHi = 0;
#pragma unroll
          for (int j = 0; j < SHORTN; j++) {
            if (j < nOwn) {
              differs |= cmd[j] != seen[j];
              const u64 wd = apply_word(cmd[j], lVerified + 1 + j);
              accLo += (unsigned)wd; accHi += (unsigned)(wd >> 32);
            }
          }
          coldLds[CD_AH_LO][tid] += (int)accLo; coldLds[CD_AH_HI][tid] += (int)accHi;
          if (differs) gfl |= RG_FLAG_STATE_MACHINE;
        }
        vm &= ~__builtin_amdgcn_ballot_w64(mine);
      }
      if (!SHORTP || vm) do {
        const int src = (int)__builtin_ctzll(vm);
        vm &= vm - 1;
        const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src), rOffB = (unsigned)__builtin_amdgcn_readlane((int)rOff, src);
        const int a = __builtin_amdgcn_readlane(lVerified, src), n = __builtin_amdgcn_readlane(nOwn, src);
        const unsigned col = (unsigned)__builtin_amdgcn_readlane(peer, src);
        const int tTid = (tid & ~63) + src;
        bool differs = false;
        unsigned accLo = 0, accHi = 0;
        for (int j = G.lane; j < n; j += 64) {
          const int i = a + 1 + j, ic = i < L ? i : L - 1;
          const u64 cmd = *lg.cmdp_at(offB, ic, col);
          const u64 seen = recBlk[rOffB + (unsigned)ic];
          differs |= cmd != seen;
          const u64 wd = apply_word(cmd, i);
          accLo += (unsigned)wd; accHi += (unsigned)(wd >> 32);
        }
        if (G.lane < n) {
          atomicAdd((unsigned *)&coldLds[CD_AH_LO][tTid], accLo);
          atomicAdd((unsigned *)&coldLds[CD_AH_HI][tTid], accHi);
        }
        if (wave_any(differs) && G.lane == src) gfl |= RG_FLAG_STATE_MACHINE;
      } while (vm);
      if (need) lVerified = r.applied;
    }
  };

  // Client Starts of the lane's group that are due up to tick `upTo` (next_event lets a group skip the tick of a
  // Start and accounts for what the Start changes; Start() itself does not look at the clock).
  auto client_starts = [&](bool on, int upTo) {
    if (!(c.startPeriod > 0)) return;
    bool startNow = on && gNextStart <= upTo && gStartSeq < c.startMax;
    if (wave_any(startNow)) do {
      // client Starts: config.go:436-451 (first connected leader) / test_test.go:697-703 (every leader)
      bool cand = startNow && valid && r.isLeader() && ((!LEAN && c.startMode == RG_START_ALL) || r.has(F_CONNECTED));
      const unsigned mAll = G.ballot(cand);
      if (startNow && mAll == 0) { do gNextStart += c.startPeriod; while (gNextStart <= upTo); }   // nobody to ask: these Starts change nothing
      if (wave_any(cand)) {
        SECTION_COUNT(w, 5);
        if (wave_any(cand && r.logLen <= r.applied)) verify_applied(cand && r.logLen <= r.applied);
        bool room = cand && r.logLen < L;
        const unsigned mRoom = G.ballot(room);
        unsigned below = (1u << peer) - 1u;
        if ((!LEAN && c.startMode == RG_START_ALL)) {
          if (startNow && mAll) gNextStart += c.startPeriod;
          int seq = gStartSeq + __popc(mRoom & below);
          bool tried = cand && seq < c.startMax;
          if (tried) {
            u64 cmd = stream_hash(key, STREAM_CMD, 0, (unsigned)seq);
            const int at = start_cmd(r, lg, sn, peer, L, cmd, gfl, w);
            if (at >= 0) RING_PUSH(at, r.term, cmd);
          }
          int acc = __popc(mRoom);
          int left = c.startMax - gStartSeq;
          if (mAll) gStartSeq += acc < left ? acc : (left > 0 ? left : 0);
        } else {
          // Peers without room are tried (and flagged) until one accepts.  Nothing else happens to the group between
          // the Starts that are due, so the peer that accepts takes them all, as far as its log has room.
          const int wIdx = mRoom ? (__ffs((int)mRoom) - 1) : NONE;
          int due = 0;
          if (startNow && mAll) { do { gNextStart += c.startPeriod; due++; } while (gNextStart <= upTo && gStartSeq + due < c.startMax); }
          int take = (cand && peer == wIdx) ? (due < L - r.logLen ? due : L - r.logLen) : 0;
          take = G.shfl(take, wIdx < NONE ? wIdx : 0);
          if (wIdx == NONE) take = 0;
          if (startNow && mAll && wIdx < NONE && take < due) gNextStart -= (due - take) * c.startPeriod;   // the rest goes round again
          if (cand && peer < wIdx) (void)start_cmd(r, lg, sn, peer, L, 0, gfl, w);            // full log: refused and flagged
          const bool lead = cand && peer == wIdx && take > 0;
          if (wave_any(lead && r.commit >= r.logLen)) {        // a commitIndex that ran past the log: entry by entry
            if (lead) {
              for (int j = 0; j < take; j++) {
                u64 cmd = stream_hash(key, STREAM_CMD, 0, (unsigned)(gStartSeq + j));
                const int at = start_cmd(r, lg, sn, peer, L, cmd, gfl, w);
                if (at >= 0) RING_PUSH(at, r.term, cmd);
              }
            }
          } else {
            // The group's lanes share the entries out: each works out its commands and writes them into the leader's
            // column (and the last RING of them into the leader's ring); the leader then does the bookkeeping of
            // `take` Start()s at once.  With commitIndex below the first new entry, persist() after the first
            // Start has nothing new to count, and no entry is the one at commitIndex.
            const int wSafe = wIdx < NONE ? wIdx : 0;
            const int idx0 = G.shfl(r.logLen, wSafe), termW = G.shfl(r.term, wSafe);
            const bool help = startNow && mAll != 0 && valid && take > 0;
            const int lt = tid - peer + wSafe;
            for (int j = peer; wave_any(help && j < take); j += P) {
              if (help && j < take) {
                const u64 cmd = stream_hash(key, STREAM_CMD, 0, (unsigned)(gStartSeq + j));
                const int at = idx0 + j;
                *lg.termp(at, wSafe) = termW;
                *lg.cmdp(at, wSafe) = cmd;
                if (take - j <= RING) { const int q = at & (RING - 1); ringIdx[q][lt] = at; ringTerm[q][lt] = termW; ringCmd[q][lt] = cmd; }
              }
            }
            if (lead) {
              if (idx0 < r.snapSync) r.snapSync = idx0;
              r.logLen = idx0 + take;
              r.cd(CD_LASTTERM) = r.term;
              persist(r, L, w);
              if (take > 1) ctr_add(w, RG_C_PERSISTS, (unsigned)(take - 1));
              ctr_add(w, RG_C_ENTRIES_STARTED, (unsigned)take);
            }
          }
          if (startNow && mAll) gStartSeq += take;
        }
        storesPending = true;
      }
      startNow = on && gNextStart <= upTo && gStartSeq < c.startMax;
    } while (wave_any(startNow));
  };

  DSTAMP(2);
  int tg;
  if constexpr (DUE) {
    // The wake word is the tick the group's last span was going to run next, so the launches continue exactly as one long
    // launch would have; only a group the host has touched since (wake 0) has to look again.
    const int wk = valid ? dueWake : 1;
    tg = wk > t0 ? wk : t0;
    if (wave_any(wk == 0)) { const int ne = next_event(t0); if (wk == 0) tg = ne; }
  } else tg = next_event(t0);
  if (!valid) tg = 0x7fffffff;
  if constexpr (DUE) {
    const int len0 = r.logLen;               // (nothing above appends to a log: set-up only)
#pragma unroll
    for (int q = 0; q < RING; q++) { const int i = len0 - 1 - q; if (pfLive && i >= 0) RING_PUSH(i, pfTerm[q], pfCmd[q]); }
  }
  DSTAMP(3);
  if (wave_any(tg < tEnd)) do {
    const bool run = tg < tEnd;            // the lane's group runs its tick tg in this pass
    SECTION_COUNT(w, 14);
    if (wave_any(run && G.peer == 0)) { if (run && G.peer == 0 && G.inWave) SECTION_COUNT_LANE(w, 15); }
    const int t = tg;
    const int now = t * RG_TICK_MS;
    STAMP_INIT;
    if (run) catch_up(t);
    STAMP(9);
    // ------------------------------------------------------------ 1. control
    const bool partNow = c.partPeriod > 0 && run && t == gPartNext;
    const bool churnNow = (!LEAN && c.churnPeriod > 0) && run && t == gChurnNext;
    // a Start of a tick the group skipped precedes this tick's churn and schedule events, which change who would take it
    // (otherwise one call below does both; groups without such an event in this tick run theirs up to this tick at the
    // same time).  A reshuffle of the partition sides changes nothing a Start looks at -- who leads, who is connected to
    // the tester -- and needs no such call.
    bool schedNow = false;
    if constexpr (EXT) schedNow = H.schedN > 0 && run && valid && gSchedPos < H.schedN && H.sched[gSchedPos].tick <= t;
    if (((!LEAN && c.churnPeriod > 0) || EXT) && wave_any(churnNow || schedNow)) client_starts(run, (churnNow || schedNow) ? t - 1 : t);
    STAMP(12);
    if constexpr (EXT) {
      // ---- 0. the scenario schedule: the tester's verbs, by every group on its own peers (config.go:99-278, 302-332, 430-477)
      if (wave_any(schedNow)) {
        storesPending = true;
        verify_applied(schedNow);              // nothing unchecked is left in a log an op may rewrite or replay
        bool due = schedNow;
        while (wave_any(due)) {
          const int pos = due ? gSchedPos : 0;
          const rg_sched_op op = H.sched[pos];
          const int kind = due ? op.op : 0;
          int regs = gRegs;
          auto resolve = [&](int sel) -> int {   // -1: nobody, -2: everybody
            if (sel & 0x100) { const int base = (regs >> (4 * ((sel >> 4) & 3))) & 15; return base == 15 ? -1 : (base + (sel & 15)) % P; }
            return sel == RG_SEL_ALL ? -2 : sel;
          };
          const int tgt = resolve(op.a);
          const bool me = due && (tgt == peer || tgt == -2);
          if (kind == RG_OP_CONNECT && me) r.flags |= F_CONNECTED;
          if (kind == RG_OP_DISCONNECT && me) r.flags &= ~F_CONNECTED;
          if (wave_any(kind == RG_OP_CONNECT_IF_FEWER)) {
            const int up = __popc(G.ballot(valid && r.has(F_CONNECTED)));
            if (kind == RG_OP_CONNECT_IF_FEWER && me && up < op.b) r.flags |= F_CONNECTED;
          }
          if (kind == RG_OP_CRASH && me) r.crash();
          if (kind == RG_OP_KILL && me) r.kill();
          const int alive = __popc(G.ballot(valid && !r.has(F_DEAD)));
          const bool again = kind == RG_OP_RESTART_IF_FEWER && me && r.has(F_DEAD) && alive < op.b;
          if ((kind == RG_OP_RESTART && me) || again) {                    // start1: crash1, then Make + readPersist
            r.crash();
            make_peer(r, lg, sn, peer, L, key, now, w);
            lVerified = 0;
#pragma unroll
            for (int q = 0; q < RING; q++) ringIdx[q][tid] = -1;
            if (again) r.flags |= F_CONNECTED;
          }
          const bool isStart = kind == RG_OP_START || kind == RG_OP_START_ANY || kind == RG_OP_START_ALL;
          int val = 15;
          if (wave_any(isStart)) {
            const bool ask = isStart && valid && !r.has(F_DEAD) &&
                             (kind == RG_OP_START ? tgt == peer : kind == RG_OP_START_ANY ? r.has(F_CONNECTED) : true);
            bool go = ask;
            if (kind == RG_OP_START_ANY) {       // leaders with a full log refuse (and are flagged) until one accepts
              const unsigned mRoom = G.ballot(ask && r.isLeader() && r.logLen < L);
              const int wIdx = mRoom ? (__ffs((int)mRoom) - 1) : NONE;
              go = ask && r.isLeader() && peer <= wIdx;
            }
            int at = -1;
            if (go) {
              const u64 cmd = stream_hash(key, STREAM_CMD, 1, (unsigned)op.b);
              at = start_cmd(r, lg, sn, peer, L, cmd, gfl, w);
              if (at >= 0) RING_PUSH(at, r.term, cmd);
            }
            const unsigned accM = G.ballot(at >= 0);
            val = accM ? 31 - __clz((int)accM) : 15;
          }
          if (wave_any(kind == RG_OP_PICK_LEADER)) {
            const bool lead = kind == RG_OP_PICK_LEADER && valid && r.has(F_CONNECTED) && r.isLeader();
            const int tmax = G.gmax(lead ? r.term : -1);
            const unsigned m = G.ballot(lead && r.term == tmax);
            if (kind == RG_OP_PICK_LEADER) val = m ? (__ffs((int)m) - 1) : 15;
          }
          if (kind == RG_OP_PICK_RANDOM) val = (int)((unsigned)(stream_hash(key, STREAM_SCHED, (unsigned)pos, (unsigned)op.b) >> 32) % (unsigned)P);
          if (wave_any(kind == RG_OP_PICK_OTHER)) {
            const int base = resolve(op.b);
            const unsigned okm = G.ballot(kind == RG_OP_PICK_OTHER && valid && r.has(F_CONNECTED) && !r.has(F_DEAD) && peer != base);
            const int from = base < 0 ? 0 : base + 1;
            const unsigned hi = from < P ? (okm >> from) : 0u;
            if (kind == RG_OP_PICK_OTHER) val = hi ? (from + __ffs((int)hi) - 1) : (okm ? (__ffs((int)okm) - 1) : 15);
          }
          if (kind == RG_OP_START_ALL || kind == RG_OP_PICK_LEADER || kind == RG_OP_PICK_RANDOM || kind == RG_OP_PICK_OTHER) {
            const int sh = 4 * (op.a & 3);
            gRegs = (regs & ~(15 << sh)) | (val << sh);
          }
          if (due) gSchedPos = pos + 1;
          due = due && gSchedPos < H.schedN && H.sched[gSchedPos].tick <= t;
        }
      }
    }
    if (partNow) {
      // one hash per reshuffle: bit p = side of peer p, the upper half scaled to the jitter
      const u64 h = stream_hash(key, STREAM_PART, (unsigned)gPartEpoch, 0);
      r.set(F_SIDE, (((unsigned)h >> peer) & 1u) != 0);
      gPartNext = t + c.partPeriod + (int)(((u64)(unsigned)(h >> 32) * (u64)(unsigned)c.partJitter) >> 32);
      gPartEpoch++;
    }
    if ((!LEAN && c.churnPeriod > 0) && wave_any(churnNow)) { storesPending = true; verify_applied(churnNow && r.has(F_DEAD)); }   // a restart replays the log
    if (churnNow) {                                                  // test_test.go:893-919
      gChurnNext = t + c.churnPeriod;
      unsigned it = (unsigned)(t / c.churnPeriod);
      u64 h1 = stream_hash(key, STREAM_CHURN, it, 0), h2 = stream_hash(key, STREAM_CHURN, it, 1), h3 = stream_hash(key, STREAM_CHURN, it, 2);
      if ((int)((unsigned)(h1 & 0xFFFFF) % 1000u) < c.churnDisc && peer == (int)((h1 >> 32) % (u64)P)) r.flags &= ~F_CONNECTED;
      if ((int)((unsigned)(h2 & 0xFFFFF) % 1000u) < c.churnRestart && valid && peer == (int)((h2 >> 32) % (u64)P)) {
        if (r.has(F_DEAD)) {
          make_peer(r, lg, sn, peer, L, key, now, w);
          lVerified = 0;                                       // lastApplied starts again from 0
#pragma unroll
          for (int q = 0; q < RING; q++) ringIdx[q][tid] = -1;     // the log may have been rebuilt from an imported blob
        }
        r.flags |= F_CONNECTED;
      }
      if ((int)((unsigned)(h3 & 0xFFFFF) % 1000u) < c.churnCrash && peer == (int)((h3 >> 32) % (u64)P) && !r.has(F_DEAD)) r.crash();
    }
    STAMP(10);
    client_starts(run, t);

    STAMP(0);
    // ------------------------------------------------------------ 2. actors
    {
      const bool live = run && valid && !r.has(F_DEAD);
      const bool elDue = live && r.elNext <= now;
      const bool monDue = live && r.has(F_MONALIVE) && r.monNext <= now;
      int start = 0, myPos = 0;
      bool pend = false, specEl = false, specMon = false, pendLocal = false;
      if (elDue) {
        int st = r.state();
        if (st == RG_FOLLOWER) {
          if (now - r.timer > r.timeout) { pend = true; pendLocal = true; }
          else r.elNext = now + RG_ELECTION_PERIOD_MS + sleep_slack(key, peerL, now);   // nothing later in this tick can make it fire
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
          r.elNext = now + RG_ELECTION_PERIOD_MS + sleep_slack(key, peerL, now);
          pend = false;
        }
      }
      STAMP(1);
      if (wave_any(pend)) { start = turn_start(key, t, P); myPos = peer - start + (peer < start ? P : 0); }

      if (wave_any(pend)) do {
        SECTION_COUNT(w, 0);
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
            r.elNext = now + RG_ELECTION_PERIOD_MS + sleep_slack(key, peerL, now);
          } else if (st == RG_CANDIDATE) {
            wantRV = true;
          } else {
            r.elNext = now + RG_ELECTION_PERIOD_MS + RG_ELECTION_LEADER_EXTRA_MS + sleep_slack(key, peerL, now);
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
            persist(r, L, w);                                       // raft.go:576
            aLli = r.commit;
            aLlt = r.cd(CD_CTERM);                                  // logs[commitIndex].Term, raft.go:583
            ctr_add(w, RG_C_PERSISTS, 1);                           // raft.go:589: the same state once more
            ctr_add(w, RG_C_RV_ROUNDS, 1);
          }
          const bool inR = G.ballot(wantRV) != 0;
          const int T = G.shfl(r.term, sIdx), lli = G.shfl(aLli, sIdx), llt = G.shfl(aLlt, sIdx);
          const unsigned seq = (unsigned)r.cold[CD_ROUNDSEQ * 256 + (sIdx - peer)];   // the sender's word, read in place
          const unsigned fS = G.shflu(r.flags, sIdx);
          if (wantRV) r.cd(CD_ROUNDSEQ) = (int)(seq + 1u);
          const bool tgt = inR && valid && peer != s;
          const bool reach = tgt && (fS & F_CONNECTED) && r.has(F_CONNECTED) && ((fS ^ r.flags) & F_SIDE) == 0 && r.served();
          int cost = 0;
          const int outW = tgt ? net_outcome(net, key, reach, s, peer, seq, cost) : 0;
          const int out = outW & 3;
          VoteDecision d; d.rTerm = 0; d.touch = d.raise = d.grant = false;
          if (out != 0) {
            if (r.has(F_DEAD)) d.rTerm = T;       // a Kill()ed peer: the shared reply stays as it is (raft.go:203, 585-588)
            else d = decide_vote(r, r.cd(CD_CTERM), T, s, lli, llt);
          }
          const unsigned sdm = G.ballot(out == 2 && d.rTerm > T);      // raft.go:622
          const int b = sdm ? (__ffs((int)sdm) - 1) : NONE;
          const bool called = tgt && peer <= b, ran = out != 0 && peer <= b, ok = out == 2 && peer <= b;
          if (ran) {
            const bool wasLead = r.state() == RG_LEADER;
            commit_vote(r, d, T, s, now, lg, sn, peer, L, w);
            if (wasLead && d.touch && myPos > sPos && (specEl || specMon)) { pend = true; specEl = false; specMon = false; }
          }
          const unsigned calledM = G.ballot(called), ranM = G.ballot(ran), okM = G.ballot(ok), grantM = G.ballot(ok && d.grant);
          if constexpr (EXT) atomicAdd((unsigned *)&coldLds[CD_N][tid], called ? (unsigned)RG_VOTE_ARGS_BYTES + ((ran && (outW & NET_WIRE)) ? (unsigned)RG_VOTE_REPLY_BYTES : 0u) : 0u);
          const int busy = G.gsum(called ? cost : 0);
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
              if (r.cd(CD_LASTTERM) == r.term) { while (f > 0 && *lg.termp(f - 1, peer) == r.term) f--; }
              r.cd(CD_FIRSTOWN) = f;
              ctr_add(w, RG_C_ELECTIONS_WON, 1);
            } else {
              r.flags |= F_RESTART;
            }
            r.elNext = now + busy + RG_ELECTION_PERIOD_MS + sleep_slack(key, peerL, now);
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
                if (N > r.commit && N >= r.cd(CD_FIRSTOWN)) {
                  ctr_add(w, RG_C_ENTRIES_COMMITTED, (unsigned)(N - r.commit));
                  r.commit = N;
                  r.cd(CD_CTERM) = r.term;                                  // N >= firstOwn
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
          const int lastTermS = r.cold[CD_LASTTERM * 256 + (sIdx - peer)], firstOwnS = r.cold[CD_FIRSTOWN * 256 + (sIdx - peer)];   // the sender's words, read in place
          const int cTermS = r.cold[CD_CTERM * 256 + (sIdx - peer)];
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
          const bool reach = tgt2 && (fS & F_CONNECTED) && r.has(F_CONNECTED) && ((fS ^ r.flags) & F_SIDE) == 0 && r.served();
          int cost = 0;
          const int outW = tgt2 ? net_outcome(net, key, reach, s, peer, seq, cost) : 0;
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
            // (all LDS words this may need are read at once, whether needed or not: one trip through the LDS pipe instead of three in a row)
            const int pq = prev & (RING - 1);
            const int myLast = r.cd(CD_LASTTERM), sIdxAt = ringIdx[pq][sTid], sTermAt = ringTerm[pq][sTid], mIdxAt = ringIdx[pq][tid], mTermAt = ringTerm[pq][tid];
            if (prev == r.logLen - 1) myPrevTerm = myLast; else needM = prev < r.logLen - 1;
            if (needS && sIdxAt == prev) { prevTerm = sTermAt; needS = false; }
            if (needM && mIdxAt == prev) { myPrevTerm = mTermAt; needM = false; }
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
          // (a Kill()ed peer leaves the reply as the leader made it, raft.go:273, 456-461: {0, false, 0, 0})
          if (out != 0 && !r.has(F_DEAD)) d = decide_append<false>(r, lg, peer, T, prev, prevTerm, myPrevTerm);
          {   // raft.go:308-313, the walk back to the first index of the conflicting term: by the wave, 64 entries a step
            u64 cm = __builtin_amdgcn_ballot_w64(out == 2 && d.ci == RG_CI_PENDING);   // only a reply that arrives carries it
            if (cm) {
              SYNC_STORES();
              do {
                const int src = (int)__builtin_ctzll(cm);
                cm &= cm - 1;
                const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src), col = (unsigned)__builtin_amdgcn_readlane(peer, src);
                const int ct = __builtin_amdgcn_readlane(d.ct, src);
                int base = __builtin_amdgcn_readlane(prev, src) - 1, ci;
                for (;;) {
                  const int ix = base - G.lane;
                  const bool stop = ix < 0 || *lg.termp_at(offB, ix, col) != ct;
                  const u64 sm = __builtin_amdgcn_ballot_w64(stop);
                  if (sm) { ci = base - (int)__builtin_ctzll(sm) + 1; break; }
                  base -= 64;
                }
                if (G.lane == src) d.ci = ci;
              } while (cm);
            }
          }
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
          // entries at or below the receiver's apply cursor are about to be rewritten: check them first
          if (wave_any(ran && d.touch && d.match && nx <= r.applied)) verify_applied(ran && d.touch && d.match && nx <= r.applied);
          {
            u64 hm = __builtin_amdgcn_ballot_w64(viaHbm);
            const bool scan = ok && b != peer && d.touch && !d.match;
            if (hm) SECTION_COUNT(w, 10);
            if (wave_any(scan)) SECTION_COUNT(w, 11);
            if (hm || wave_any(scan)) SYNC_STORES();
            if constexpr (SHORTP) {
              const bool mine = viaHbm && k <= SHORTN;
              if (mine) {
                int tm[SHORTN]; u64 cm[SHORTN];
#pragma unroll
                for (int j = 0; j < SHORTN; j++) { const int i = nx + (j < k ? j : k - 1); tm[j] = *lg.termp(i, sIdx); cm[j] = *lg.cmdp(i, sIdx); }
#pragma unroll
                for (int j = 0; j < SHORTN; j++) {
                  if (j < k) { const int i = nx + j; *lg.termp(i, peer) = tm[j]; *lg.cmdp(i, peer) = cm[j]; RING_PUSH(i, tm[j], cm[j]); }
                }
              }
              hm &= ~__builtin_amdgcn_ballot_w64(mine);
            }
            while (hm) {
              const int src = (int)__builtin_ctzll(hm);
              hm &= hm - 1;
              const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src);
              const int nxB = __builtin_amdgcn_readlane(nx, src), lenB = __builtin_amdgcn_readlane(lenS, src);
              const unsigned sCol = (unsigned)__builtin_amdgcn_readlane(sIdx, src), tCol = (unsigned)__builtin_amdgcn_readlane(peer, src);
              const int tTid = (tid & ~63) + src;
              for (int i = nxB + G.lane; i < lenB; i += 64) {
                const int tm = *lg.termp_at(offB, i, sCol);
                const u64 cm = *lg.cmdp_at(offB, i, sCol);
                *lg.termp_at(offB, i, tCol) = tm;
                *lg.cmdp_at(offB, i, tCol) = cm;
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
              r.cd(CD_LASTTERM) = k > 0 ? lastTermS : prevTerm;
              const int oldCommit = r.commit;
              if (commitS > r.commit) r.commit = (r.logLen - 1 < commitS) ? r.logLen - 1 : commitS;
              if (r.commit > r.logLen - 1) gfl |= RG_FLAG_REF_PANIC;
              if (r.commit != oldCommit || nx <= r.commit) {          // the entry at commitIndex is new or was rewritten
                // (the leader's word only stands for an entry this very request carried: this reference does not keep
                // logs that agree at PrevLogIndex equal below it, raft.go:218-219 elects on commitIndex)
                if (r.commit == lenS - 1) r.cd(CD_CTERM) = r.cd(CD_LASTTERM);
                else if (nx <= r.commit && r.commit == commitS && commitS < lenS) r.cd(CD_CTERM) = cTermS;
                else { wave_fence(); r.cd(CD_CTERM) = *lg.termp(r.commit < L ? r.commit : L - 1, peer); WAIT_LOADS(); }
              }
              persist(r, L, w);
              ctr_add(w, RG_C_ENTRIES_REPLICATED, (unsigned)k);
            } else {
              ctr_add(w, RG_C_AE_CONFLICTS, 1);
            }
            if (wasLead && myPos > sPos && (specEl || specMon)) { pend = true; specEl = false; specMon = false; }
          }
          // the leader's handling of each reply, done by the replying lane
          bool matchMoved = false;
          if (ok && b != peer) {
            if (d.match) { matchMoved = cMatch != lenS - 1; cNext = lenS; cMatch = lenS - 1; matchRow(sIdx * PP + peer) = cMatch; }
            else cNext = d.ci;                                       // raft.go:493-505; the walk over the leader's log follows
            nextRow(sIdx * PP + peer) = cNext;
          }
          {   // raft.go:495-502: the leader looks for its last entry of the conflicting term, by the wave
            const bool walk = ok && b != peer && !d.match && d.ct >= 0;
            u64 wm = __builtin_amdgcn_ballot_w64(walk);
            while (wm) {
              const int src = (int)__builtin_ctzll(wm);
              wm &= wm - 1;
              const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src), col = (unsigned)__builtin_amdgcn_readlane(sIdx, src);
              const int ct = __builtin_amdgcn_readlane(d.ct, src);
              int base = __builtin_amdgcn_readlane(nx, src) - 1, hit, hitTerm;
              for (;;) {                                 // ends at index 0 at the latest: its term 0 is not above ct
                const int ix = base - G.lane;
                const int tmv = ix >= 0 ? *lg.termp_at(offB, ix, col) : 0;
                const u64 sm = __builtin_amdgcn_ballot_w64(tmv <= ct);
                if (sm) { const int f = (int)__builtin_ctzll(sm); hit = base - f; hitTerm = __builtin_amdgcn_readlane(tmv, f); break; }
                base -= 64;
              }
              if (G.lane == src && hitTerm == ct) { cNext = hit + 1; nextRow(sIdx * PP + peer) = cNext; }
            }
          }
          const unsigned calledM = G.ballot(called), ranM = G.ballot(ran), okM = G.ballot(ok);
          if constexpr (EXT) {   // wire bytes: summed per target lane in LDS words of its own, added to the counters when the launch ends
            atomicAdd((unsigned *)&coldLds[CD_N][tid], called ? (unsigned)RG_APPEND_ARGS_BYTES + ((ran && (outW & NET_WIRE)) ? (unsigned)RG_APPEND_REPLY_BYTES : 0u) : 0u);
            atomicAdd((unsigned *)&coldLds[CD_N + 1][tid], (called && k > 0) ? (unsigned)k : 0u);
          }
          const int busy = G.gsum(called ? cost : 0);
          int rTermB = 0;
          if (wave_any(b < NONE)) rTermB = G.shfl(d.rTerm, b < NONE ? b : 0);
          // Would the commit rule (raft.go:400-416) move commitIndex with the matchIndex row as this round leaves it?
          // Nothing but the leader's next round changes the row, so if not, the evaluation that follows the round's
          // waits finds nothing to do and need not wake the group.
          int nAfter = 0;
          bool ruleOpen = false;
          {
            const bool folR = inR && valid && peer != s;
            ruleOpen = G.ballot(folR && cOwner != s) != 0;       // a follower the round did not reach: decided at the evaluation
            // (the rule was evaluated right before the round: with no matchIndex moved it finds what it found then)
            if (wave_any(matchMoved)) {
              const int mv = folR ? cMatch : -1;
              int ge = 0;
#pragma unroll
              for (int u = 0; u < P; u++) ge += G.shfl(mv, u) >= mv ? 1 : 0;
              nAfter = G.gmax((folR && ge >= P / 2) ? mv : 0);
            }
          }
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
            { int N = nAfter; if (N > r.logLen - 1) N = r.logLen - 1; if (ruleOpen || (N > r.commit && N >= r.cd(CD_FIRSTOWN))) r.flags |= F_CDIRTY; }
            if (busy > 0) { r.monNext = now + busy; monAct = false; }   // the loop goes on once the waits are over
          }
          storesPending = true;
          STAMP(8);
        }
        if (isS) pend = false;
      } while (wave_any(pend));
      if (specEl) r.elNext = now + RG_ELECTION_PERIOD_MS + RG_ELECTION_LEADER_EXTRA_MS + sleep_slack(key, peerL, now);
      if (specMon) r.monNext = now + RG_MONITOR_POLL_MS;
    }

    STAMP(2);
    // ------------------------------------------------------------ 3. apply, raft.go:676-686 + config.go:170-194
    // The reference's tester sees peers 0..P-1 deliver in turn: peer p checks its entries up to the newest index
    // recorded before its turn (pm = the larger of the record's end and the commitIndex of every lower peer)
    // against the record, and records the rest.  Ranges of different peers that are recorded never overlap.  Here
    // the cursor moves and new indices are recorded at once; the checks and the digest follow in verify_applied.
    {
      const bool act = run && valid && !r.has(F_DEAD) && r.commit > r.applied;
      if (wave_any(act)) {
        SECTION_COUNT(w, 4);
        const int cA = act ? r.commit : 0;
        const int applied = r.applied;
        int pm = gRecMax, newMax = gRecMax;
        if (wave_any(cA > gRecMax)) {
#pragma unroll
          for (int u = 0; u < P; u++) {
            const int v = G.shfl(cA, u);
            if (u < peer) pm = v > pm ? v : pm;
            newMax = v > newMax ? v : newMax;
          }
        }
        const int recLo = applied > pm ? applied : pm;
        const int nRec = (act && r.commit > recLo) ? r.commit - recLo : 0;         // records (recLo, commit]
        if (nRec > 0 && applied > pm) gfl |= RG_FLAG_APPLY_ORDER;                  // config.go:184-187
        if (act && r.commit >= r.logLen) gfl |= RG_FLAG_REF_PANIC;                 // raft.go:679 would index past the log
        if (wave_any(nRec > 0)) {
          bool slow = nRec > RING;
#pragma unroll
          for (int q = 0; q < RING; q++) {
            const int want = recLo + 1 + ((q - recLo - 1) & (RING - 1));
            if (want <= r.commit && nRec > 0 && ringIdx[q][tid] != want) slow = true;
          }
          u64 sm = __builtin_amdgcn_ballot_w64(slow);
          if constexpr (SHORTP) {
            const bool mine = slow && nRec <= SHORTN;
            if (wave_any(mine)) {
              SYNC_STORES();
              if (mine) {
                u64 cmd[SHORTN];
#pragma unroll
                for (int j = 0; j < SHORTN; j++) { const int i = recLo + 1 + (j < nRec ? j : nRec - 1); cmd[j] = *lg.cmdp(i < L ? i : L - 1, peer); }
#pragma unroll
                for (int j = 0; j < SHORTN; j++) { if (j < nRec) { const int i = recLo + 1 + j; rec(i < L ? i : L - 1) = cmd[j]; } }
              }
              sm &= ~__builtin_amdgcn_ballot_w64(mine);
            }
          }
          if (nRec > 0 && !slow) {
            for (int j = 0; j < nRec; j++) {
              const int i = recLo + 1 + j, q = i & (RING - 1);
              const u64 cmd = ringCmd[q][tid];
              rec(i < L ? i : L - 1) = cmd;
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
              const unsigned col = (unsigned)__builtin_amdgcn_readlane(peer, src);
              for (int j = G.lane; j < n; j += 64) {
                const int i = lo + 1 + j, ic = i < L ? i : L - 1;
                const u64 cmd = *lg.cmdp_at(offB, ic, col);
                recBlk[rOffB + (unsigned)ic] = cmd;
              }
            } while (sm);
          }
          gRecMax = newMax;
          storesPending = true;
        }
        if constexpr (EXT) {
          // applyCh (raft.go:680-684): every ApplyMsg goes into the peer's ring at apply time, with the command the log
          // holds then; head counts the messages the slot's peers have emitted since the engine was created or reset
          // (a restarted peer replays from index 1 into the same ring).  By the whole wave, one entry per lane.
          u64 am = H.applyCap > 0 ? __builtin_amdgcn_ballot_w64(act) : 0ull;
          if (am) {
            SYNC_STORES();
            const unsigned dense = (unsigned)(g * P + peer);       // the ring's index of this peer
            unsigned head = 0;
            if (act) { head = H.applyHead[gl]; WAIT_LOADS(); }
            const int nOwn = act ? r.commit - applied : 0;
            do {
              const int src = (int)__builtin_ctzll(am);
              am &= am - 1;
              const unsigned offB = (unsigned)__builtin_amdgcn_readlane((int)lg.off, src), col = (unsigned)__builtin_amdgcn_readlane(peer, src);
              const int lo = __builtin_amdgcn_readlane(applied, src), n = __builtin_amdgcn_readlane(nOwn, src);
              const unsigned hd = (unsigned)__builtin_amdgcn_readlane((int)head, src), dn = (unsigned)__builtin_amdgcn_readlane((int)dense, src);
              rg_apply_msg *const ring = H.applyRing + (size_t)dn * (size_t)H.applyCap;
              // (of a burst longer than the ring only the last applyCap messages stay: the earlier ones would land in the
              // same slots, and lanes that write one slot do so in no particular order)
              for (int j = (n > H.applyCap ? n - H.applyCap : 0) + G.lane; j < n; j += 64) {
                const int i = lo + 1 + j, ic = i < L ? i : L - 1;
                const u64 cmd = *lg.cmdp_at(offB, ic, col);
                rg_apply_msg m; m.command_index = i; m.command = cmd;
                ring[(hd + (unsigned)j) % (unsigned)H.applyCap] = m;
              }
            } while (am);
            if (act) H.applyHead[gl] = head + (unsigned)nOwn;
            storesPending = true;
          }
        }
        if (act) {
          ctr_add(w, RG_C_ENTRIES_APPLIED, (unsigned)(r.commit - applied));
          r.applied = r.commit;
        }
        // keep the unchecked stretch short, so that a check forced by a rewrite (in the middle of a round) is one step
        if (wave_any(act && r.commit - lVerified >= RG_VERIFY_BATCH)) verify_applied(act && r.commit - lVerified >= RG_VERIFY_BATCH);
      }
    }

    STAMP(3);
    // ------------------------------------------------------------ 4. one leader per term, config.go:307-320
    {
      const bool lead = run && valid && r.has(F_CONNECTED) && r.isLeader();
      const unsigned lm = G.ballot(lead);
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
    {   // this group's next tick with something to do
      const int ne = next_event(t + 1);
      if (run) tg = ne;
    }
    STAMP(11);
  } while (wave_any(tg < tEnd));
  DSTAMP(4);
  client_starts(valid, tEnd - 1);          // Starts and wake-ups the groups skipped towards the end of the launch
  DSTAMP(5);
  catch_up(tEnd);
  DSTAMP(6);
  verify_applied(valid);
  DSTAMP(7);

#undef SYNC_STORES
#undef RING_PUSH
  const int startSeqOut = gStartSeq, partEpochOut = gPartEpoch, partNextOut = gPartNext, recMaxOut = gRecMax;
  int schedPosOut = 0, regsOut = 0;
  if constexpr (EXT) { schedPosOut = gSchedPos; regsOut = gRegs; }
  if (EXT && c.acctBytes) {
    if (coldLds[CD_N][threadIdx.x]) ctr_add(w, RG_C_RPC_FIXED_BYTES, (unsigned)coldLds[CD_N][threadIdx.x]);
    if (coldLds[CD_N + 1][threadIdx.x]) ctr_add(w, RG_C_AE_ENTRIES_SENT, (unsigned)coldLds[CD_N + 1][threadIdx.x]);
  }
#undef gStartSeq
#undef gSchedPos
#undef gRegs
#undef cOwner
#undef cNext
#undef cMatch
#undef lVerified
#undef gPartNext
#undef gRecMax
#undef gNextStart
#undef gChurnNext
#undef gPartEpoch
#undef nextRow
#undef matchRow
#undef rec
#undef rOff
  if (DUE ? valid : G.inWave) r.store(*Sp, gl);
  gfl = G.gor(gfl);
  if (stOut) {                               // rounds sent in this span: how busy the group is, for the next span's dealing
    const int sent = G.gsum(valid ? r.cd(CD_ROUNDSEQ) : 0);
    if (valid && peer == 0) stOut[6 * GPBPAD + gLoc] = sent - stOut[5 * GPBPAD + gLoc];
  }
  if constexpr (EXT) {
    if (H.schedN > 0 && valid && peer == 0) { Sp->schedPos[g] = schedPosOut; Sp->schedRegs[g] = regsOut; }
  }
  if (valid && peer == 0) {
    Sp->startSeq[g] = startSeqOut; Sp->partNext[g] = partNextOut; Sp->partEpoch[g] = partEpochOut; Sp->recMax[g] = recMaxOut;
    // the group's next tick with something to do (>= tEnd), and: all lazy work done up to tEnd -- one store for the pair
    *(u64 *)&RG_WAKE(*Sp, g) = ((u64)(unsigned)tEnd << 32) | (u64)(unsigned)tg;
    if (stOut) {
      const unsigned all = (stIn ? (unsigned)stIn[4 * GPBPAD + gLoc] : Sp->gflags[g]) | gfl;
      stOut[gLoc] = startSeqOut; stOut[GPBPAD + gLoc] = partEpochOut; stOut[2 * GPBPAD + gLoc] = partNextOut; stOut[3 * GPBPAD + gLoc] = recMaxOut;
      stOut[4 * GPBPAD + gLoc] = (int)all;
      if (gfl) Sp->gflags[g] = all;
    } else if (gfl) Sp->gflags[g] |= gfl;
  }
  DSTAMP(8);
#ifdef RG_DSTAMPS
  {
    const unsigned ng = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid && peer == 0));
    const unsigned tot = (unsigned)(__builtin_readcyclecounter() - _dstamp0);
    if (G.lane == 0) {
      atomicAdd(&w->c[40 + 9], 1u); atomicAdd(&w->c[40 + 10], ng); atomicAdd(&w->c[40 + 15], (unsigned)(__builtin_amdgcn_s_memrealtime() - _dreal0));
#ifndef RG_DST_T1
#define RG_DST_T1 60000u
#define RG_DST_T2 120000u
#define RG_DST_T3 200000u
#endif
      if (tot > RG_DST_T1) atomicAdd(&w->c[40 + 12], 1u);
      if (tot > RG_DST_T2) atomicAdd(&w->c[40 + 13], 1u);
      if (tot > RG_DST_T3) atomicAdd(&w->c[40 + 14], 1u);
    }
  }
#endif
}

// The tick kernel.  Plain launch (c.relaySets == 0): block b runs set b through all nticks.
//
// Relay launch: every wave costs about the same time whatever its groups do, so a launch of 1 1/3 times the
// groups the chip holds would take as long as one of 2 times.  Instead the work is cut into items (set, chunk of
// c.relayChunk ticks), one block each, and a block takes its item from a counter when it starts, in chunk-major
// order: as blocks finish, the dispatcher starts the next ones, so all SIMDs stay busy until the last items.  Item
// (set, k) needs the state item (set, k - 1) stored, possibly by a block on another XCD: the finisher publishes
// relay[1 + set] = k with agent-scope release after a fence, the taker spins on it with acquire.  A block only ever
// waits for an item drawn before its own, i.e. by a block that is already running, so the smallest unfinished item
// never waits and every block ends.  A chunk boundary is what a launch boundary is: the result does not depend on
// the cut.  Adjacent sets share 128-byte lines of the per-group arrays, and a line this XCD's L2 holds with bytes
// of its own to write back is not known to be refetched for its other bytes: the per-group words travel through
// a staging area with whole lines per set instead (the host lays out relay launches so that no other array has
// a line with two sets in it).
#ifdef RG_SPEC_P
// A build for one engine's configuration (rg_specialize compiles this file at run time): what rg_create fixed is a
// compile-time constant here, so it costs no scalar register and folds into the arithmetic around it.
__device__ __forceinline__ void spec_constants(DevCfg &c) {
  c.P = RG_SPEC_P; c.L = RG_SPEC_L; c.rowBytes = RG_SPEC_ROWBYTES; c.cmdOff = RG_SPEC_CMDOFF; c.cols = RG_SPEC_COLS; c.groupBytes = RG_SPEC_GROUPBYTES;
  c.reqDrop = RG_SPEC_REQDROP; c.replyDrop = RG_SPEC_REPLYDROP; c.replyDelay = RG_SPEC_REPLYDELAY;
  c.startPeriod = RG_SPEC_STARTPERIOD; c.startFirst = RG_SPEC_STARTFIRST; c.startMax = RG_SPEC_STARTMAX; c.startMode = RG_SPEC_STARTMODE;
  c.partPeriod = RG_SPEC_PARTPERIOD; c.partJitter = RG_SPEC_PARTJITTER;
  c.churnPeriod = RG_SPEC_CHURNPERIOD; c.churnDisc = RG_SPEC_CHURNDISC; c.churnRestart = RG_SPEC_CHURNRESTART; c.churnCrash = RG_SPEC_CHURNCRASH;
  c.acctBytes = RG_SPEC_ACCTBYTES;
}
#endif
template <int P, bool EXT, bool LEAN, bool COLS>
__device__ __forceinline__ void tick_launch(const DevPtrs *Sp, HotPtrs &H, DevCfg &c, int t0, int nticks) {
#ifdef RG_SPEC_P
  spec_constants(c);
#endif

  __shared__ WaveCtr sctr[4];
  __shared__ int sItem;
  const int lane = (int)(threadIdx.x & 63);
  WaveCtr *w = &sctr[threadIdx.x >> 6];
  if (lane < RG_WAVE_CTR_SLOTS) w->c[lane] = 0;
  int set = (int)blockIdx.x;
  const int *stIn = nullptr;
  int *stOut = nullptr;
  if (c.relaySets > 0) {
    if (threadIdx.x == 0) sItem = atomicAdd(H.relay, 1);
    __syncthreads();
    const int chunk = sItem / c.relaySets;
    set = sItem - chunk * c.relaySets;
    const int from = chunk * c.relayChunk;          // < nticks: the grid is exactly the items
    t0 += from;
    nticks = nticks - from < c.relayChunk ? nticks - from : c.relayChunk;
    constexpr int STAGE = 7 * ((4 * GrpN<P>::GWMAX + 31) & ~31);       // ints per set
    stOut = H.relayStage + (size_t)set * STAGE;
    if (chunk > 0) stIn = stOut;
    if (chunk > 0) {
      if (threadIdx.x == 0) {
        while (__hip_atomic_load(&H.relay[1 + set], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < chunk) __builtin_amdgcn_s_sleep(64);
      }
      __syncthreads();                               // the acquire invalidated this CU's vector cache for all its waves
    }
  }
  __syncthreads();
  tick_span<P, EXT, LEAN, COLS>(Sp, H, c, set, t0, nticks, w, stIn, stOut);
  if (c.relaySets > 0) {
    __syncthreads();                                 // every wave's stores have reached the L2; thread 0's release writes it back
    if (threadIdx.x == 0) {
      const int chunk = sItem / c.relaySets;
      __hip_atomic_store(&H.relay[1 + sItem - chunk * c.relaySets], chunk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (lane < RG_WAVE_CTR_SLOTS && w->c[lane]) atomicAdd((unsigned long long *)&Sp->counters[lane], (unsigned long long)w->c[lane]);
}

// Due launch: a launch of few ticks in which most groups have nothing to do (on the benchmark workload a group has an
// event in one tick of six).  Every span leaves the group's next event tick in `wake`; a block of this kernel looks at the
// wake words of its window of c.dueWin (<= 256) consecutive groups, one coalesced read, and runs the span for those that
// fall inside [t0, t0 + nticks) only, shared out evenly among its four waves, 64/P to a wave at most; the others are not
// touched at all -- their lazy work (wake-up chains, client Starts) waits until they next run, exactly as it does inside
// a long launch.  A span of a few ticks is bound by the instructions of its set-up, its one or two passes and its
// wind-up, which cost a wave the same whether it holds one group or twelve: hence few, full waves, about one per SIMD.
// The window's groups are at most 256 apart, so the uniform base + 32-bit lane offset addressing of tick_span holds.
template <int P, bool EXT, bool LEAN, bool COLS>
__device__ __forceinline__ void due_launch(const DevPtrs *Sp, const HotPtrs &H, const DevCfg &c, int t0, int nticks) {
  constexpr int GWMAX = GrpN<P>::GWMAX;
  __shared__ WaveCtr dctr[4];
  __shared__ int pick[256];
  __shared__ int waveDue[4];
  __shared__ int wakeOf[256];
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  WaveCtr *w = &dctr[wave];
  if (lane < RG_WAVE_CTR_SLOTS) w->c[lane] = 0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RG_NO_KERNARG_TOUCH)   // (device pass only: the host pass cannot parse the constraints)
  // The kernel's arguments are some 550 bytes that the compiler fetches where they are first used -- most of them at the
  // start of the first span, a trip to memory of their own.  One word of each of their 64-byte lines is asked for here,
  // so the lines arrive in the scalar cache while the wake words are on their way.
  int ka0, ka1, ka2, ka3, ka4, ka5, ka6, ka7, ka8;
  {
    static_assert(sizeof(DevPtrs) + sizeof(HotPtrs) + sizeof(DevCfg) >= 0x204, "the touched lines lie inside the arguments");
    const void *ka = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("s_load_dword %0, %9, 0x0\n\ts_load_dword %1, %9, 0x40\n\ts_load_dword %2, %9, 0x80\n\ts_load_dword %3, %9, 0xc0\n\t"
                 "s_load_dword %4, %9, 0x100\n\ts_load_dword %5, %9, 0x140\n\ts_load_dword %6, %9, 0x180\n\ts_load_dword %7, %9, 0x1c0\n\t"
                 "s_load_dword %8, %9, 0x200"
                 : "=&s"(ka0), "=&s"(ka1), "=&s"(ka2), "=&s"(ka3), "=&s"(ka4), "=&s"(ka5), "=&s"(ka6), "=&s"(ka7), "=&s"(ka8) : "s"(ka));
  }
#endif
  // c.dueSplit blocks share a window: block r of them takes rounds r, r + c.dueSplit, ...  In a one-tick launch of the
  // benchmark workload a window has 41 +- 6 due groups, one round of 48 for most blocks; the second block of a window runs
  // the surplus of the others at the same time instead of after it (and ends at once when there is none).
  // (the windows' first blocks come first in the grid, their further blocks after them: consecutive blocks go to different
  // XCDs in turn, and with a window's two blocks next to each other every other XCD would get nothing but second blocks)
  const int nWin = (int)gridDim.x / c.dueSplit;
  const long long first = (long long)((int)blockIdx.x % nWin) * c.dueWin;        // the window's first group
  const int share = (int)blockIdx.x / nWin;
  const int tEnd = t0 + nticks;
  const int i = (int)threadIdx.x;
  // due when the launch began: not run yet and waking inside it, or already run by another block of this launch
  bool due = false;
  if (i < c.dueWin && first + i < c.G) {
    const u64 pr = *(const u64 *)&RG_WAKE(*Sp, first + i);
    due = (int)(unsigned)pr < tEnd || (int)(unsigned)(pr >> 32) == tEnd;
    wakeOf[i] = (int)(unsigned)pr;
  }
  const u64 m = __builtin_amdgcn_ballot_w64(due);
  if (lane == 0) waveDue[wave] = (int)__builtin_popcountll(m);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RG_NO_KERNARG_TOUCH)   // (device pass only: the host pass cannot parse the constraints)
  // (the registers stay reserved until the words have arrived)
  asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(ka0), "s"(ka1), "s"(ka2), "s"(ka3), "s"(ka4), "s"(ka5), "s"(ka6), "s"(ka7), "s"(ka8));
#endif
  __syncthreads();
  int before = 0, nDue = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { const int n = waveDue[q]; before += q < wave ? n : 0; nDue += n; }
  if ((m >> lane) & 1ull) pick[before + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;   // the k-th due group of the window
  __syncthreads();
  for (int b0 = share * 4 * GWMAX; b0 < nDue; b0 += c.dueSplit * 4 * GWMAX) {
    const int nb = nDue - b0 < 4 * GWMAX ? nDue - b0 : 4 * GWMAX;  // groups of this round, shared out evenly
    const int per = (nb + 3) >> 2;
    const int lo = b0 + wave * per, hi = b0 + ((wave + 1) * per < nb ? (wave + 1) * per : nb);
    if (lo >= hi) continue;                                        // nothing left for this wave (the same for all its lanes)
    const int k = lo + lane / P;
    const bool v = lane / P < GWMAX && k < hi;
    const int off = pick[v ? k : lo];                              // lanes without a group look at the wave's first one and store nothing
    tick_span<P, EXT, LEAN, COLS, true>(Sp, H, c, (int)first, t0, nticks, w, nullptr, nullptr, off, v, wakeOf[off]);
  }
  __syncthreads();
  if (lane < RG_WAVE_CTR_SLOTS && w->c[lane]) atomicAdd((unsigned long long *)&Sp->counters[lane], (unsigned long long)w->c[lane]);
}

// (the extended instantiation carries the schedule's and the apply ring's code and is not asked to fit 4 waves per SIMD)
#ifdef RG_SPEC_P
#ifndef RG_DUE_WAVES
#define RG_DUE_WAVES 2
#endif
extern "C" __global__ __launch_bounds__(256, RG_SPEC_EXT ? 2 : RG_TICK_WAVES) void k_tick_spec(const DevPtrs *Sp, HotPtrs H, DevCfg c, int t0, int nticks) {
  tick_launch<RG_SPEC_P, RG_SPEC_EXT != 0, RG_SPEC_LEAN != 0, RG_SPEC_COLS != 0>(Sp, H, c, t0, nticks);
}
// the due launch of the same configuration (about one wave per SIMD is busy: registers are worth more to it than occupancy)
// (the pointer block comes by value here: a span of a due launch begins with a chain of dependent loads, and this takes the
// first link out of it)
extern "C" __global__ __launch_bounds__(256, RG_DUE_WAVES) void k_tick_spec_due(DevPtrs S, HotPtrs H, DevCfg c, int t0, int nticks) {
  spec_constants(c);
  due_launch<RG_SPEC_P, RG_SPEC_EXT != 0, RG_SPEC_LEAN != 0, RG_SPEC_COLS != 0>(&S, H, c, t0, nticks);
}
#else
template <int P, bool EXT, bool LEAN, bool COLS>
__global__ __launch_bounds__(256, EXT ? 2 : RG_TICK_WAVES) void k_tick(const DevPtrs *Sp, HotPtrs H, DevCfg c, int t0, int nticks) {
  tick_launch<P, EXT, LEAN, COLS>(Sp, H, c, t0, nticks);
}
// (a due launch keeps about one wave per SIMD busy: registers are worth more to it than occupancy)
#ifndef RG_DUE_WAVES
#define RG_DUE_WAVES 2
#endif
template <int P, bool EXT, bool COLS>
__global__ __launch_bounds__(256, RG_DUE_WAVES) void k_tick_due(DevPtrs S, HotPtrs H, DevCfg c, int t0, int nticks) {
  due_launch<P, EXT, false, COLS>(&S, H, c, t0, nticks);
}
#endif

#endif
