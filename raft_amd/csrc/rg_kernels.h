// rg_kernels.h -- one-thread-per-command kernels (host verbs, batched RPC handlers), cache rebuild, safety scan, digests
// (needs rg_state.h and the Grp helper of rg_tick.h)
#ifndef RG_KERNELS_H
#define RG_KERNELS_H

// ---------------------------------------------------------------- host-verb kernels
// One thread per command; commands of one launch touch distinct groups.
enum { OP_START = 0, OP_CONNECT, OP_DISCONNECT, OP_SIDE0, OP_SIDE1, OP_CRASH, OP_RESTART, OP_GETSTATE, OP_KILL };
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
    // a group that launches of a few ticks have left alone: its wake-up chains are behind, and a crash or Kill ends them
    if (cm.op == OP_CRASH || cm.op == OP_KILL) catch_up_lane(r, key, p, now / RG_TICK_MS);
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
    case OP_CRASH: r.crash(); break;
    case OP_KILL: r.kill(); break;
    case OP_RESTART: r.crash(); make_peer(r, lg, sn, p, c.L, key, now, &sctr); break;
    case OP_GETSTATE: cm.r1 = r.term; cm.r2 = r.isLeader() ? 1 : 0; break;
    }
    r.store(S, gl);
    if (cm.op != OP_GETSTATE) RG_WAKE(S, g) = 0;          // the group's next event has to be worked out again
    if (gfl) atomicOr(&S.gflags[g], gfl);
    cmds[i] = cm;
  }
  __syncthreads();
  if (threadIdx.x < RG_N_COUNTERS && sctr.c[threadIdx.x]) atomicAdd((unsigned long long *)&S.counters[threadIdx.x], (unsigned long long)sctr.c[threadIdx.x]);
}

// Start (raft.go:344-364) for a batch that names every group at most once: 16 bytes in and 16 bytes out per call, which is
// what crosses PCIe (the general command record is 40 bytes each way).
struct StartIn { int group, peer; u64 cmd; };
struct StartOut { int index, term, isLeader, pad; };
__global__ void k_start_batch(DevPtrs S, DevCfg c, int PP, const StartIn *in, StartOut *out, int n) {
  __shared__ WaveCtr sctr;
  if (threadIdx.x < RG_N_COUNTERS) sctr.c[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const StartIn q = in[i];
    const int g = q.group, p = q.peer;
    const long long gl = (long long)g * PP + p;
    const Rows lg = rows_of(S.logRows, c, g), sn = rows_of(S.snapRows, c, g);
    int coldLocal[CD_N];
    LaneT<int *> r; r.cold = coldLocal; r.cs = 1; r.load(S, gl);
    unsigned gfl = 0;
    StartOut o; o.term = r.term; o.isLeader = r.isLeader() ? 1 : 0; o.pad = 0;
    o.index = start_cmd(r, lg, sn, p, c.L, q.cmd, gfl, &sctr);
    if (o.index < 0) o.isLeader = 0;
    else {                                            // only an accepted Start changes anything
      r.store(S, gl);
      RG_WAKE(S, g) = 0;
    }
    if (gfl) atomicOr(&S.gflags[g], gfl);
    out[i] = o;
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
      VoteDecision d = decide_vote(r, r.cd(CD_CTERM), a.term, a.candidate, a.last_log_index, a.last_log_term);
      commit_vote(r, d, a.term, a.candidate, now, lg, sn, p, c.L, &sctr);
      replies[i].term = d.rTerm; replies[i].granted = d.grant ? 1 : 0;
      r.store(S, gl);
      RG_WAKE(S, g) = 0;
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
          r.cd(CD_LASTTERM) = a.n_entries > 0 ? eterms[a.entry_off + a.n_entries - 1] : a.prev_log_term;
          if (a.leader_commit > r.commit) r.commit = (r.logLen - 1 < a.leader_commit) ? r.logLen - 1 : a.leader_commit;
          if (r.commit > r.logLen - 1) gfl |= RG_FLAG_REF_PANIC;
          __threadfence();
          r.cd(CD_CTERM) = *lg.termp(r.commit < c.L ? r.commit : c.L - 1, p);
          persist(r, c.L, &sctr);
          ctr_add(&sctr, RG_C_ENTRIES_REPLICATED, (unsigned)a.n_entries);
        } else {
          rep.conflict_term = d.ct; rep.conflict_first_index = d.ci;
          ctr_add(&sctr, RG_C_AE_CONFLICTS, 1);
        }
      }
      replies[i] = rep;
      r.store(S, gl);
      RG_WAKE(S, g) = 0;
      if (gfl) atomicOr(&S.gflags[g], gfl);
    }
  }
  __syncthreads();
  if (threadIdx.x < RG_N_COUNTERS && sctr.c[threadIdx.x]) atomicAdd((unsigned long long *)&S.counters[threadIdx.x], (unsigned long long)sctr.c[threadIdx.x]);
}

// Make (raft.go:700-731 with an empty Persister, and the head of election(), raft.go:538-547) for the peer of lane gl
// at time nowTick: every per-peer word, the peer's rows of the leader matrices and, by peer 0, the group words.
// Lanes past the last group or peer hold zeros.
__device__ static inline void init_lane(const DevPtrs &S, const DevCfg &c, int PP, long long gl, int nowTick) {
  const int g = (int)(gl / PP), p = (int)(gl % PP);
  const bool peer = g < c.G && p < c.P;
  const int nowMs = peer ? nowTick * RG_TICK_MS : 0;
  int timeout = 0;
  if (peer && p != 0) {                                          // raft.go:545-547
    const u64 key = sm64(c.seedKey ^ (u64)(c.groupBase + g));
    const u64 h = stream_hash(key, STREAM_ELECTION, (unsigned)p, 0);
    timeout = (int)((unsigned)(h >> 32) % (unsigned)RG_TIMEOUT_SPAN_MS) + RG_TIMEOUT_BASE_MS;
  }
  S.term[gl] = 0; S.vote[gl] = -1; S.commit[gl] = 0; S.applied[gl] = 0; S.timer[gl] = nowMs; S.hbTimer[gl] = nowMs;
  S.elNext[gl] = nowMs; S.monNext[gl] = 0; S.roundSeq[gl] = 0; S.snapTerm[gl] = 0; S.snapVote[gl] = 0; S.snapLen[gl] = 0;
  S.lastTerm[gl] = 0; S.commitTerm[gl] = 0; S.applyHash[gl] = 0;
  S.flags[gl] = peer ? (int)F_CONNECTED : 0;
  S.logLen[gl] = peer; S.snapSync[gl] = peer; S.firstOwn[gl] = peer;
  S.timeout[gl] = timeout; S.rngCtr[gl] = peer && p != 0;
  for (int q = 0; q < PP; q++) { S.nextIdx[gl * PP + q] = 0; S.matchIdx[gl * PP + q] = 0; }
  if (p == 0) {
    S.startSeq[g] = 0; S.partEpoch[g] = 0; S.recMax[g] = 0; S.gflags[g] = 0;
    RG_WAKE(S, g) = 0; RG_SETTLED(S, g) = nowTick;
    S.schedRegs[g] = 0xFFFF;                                   // four empty registers; the schedule's position is the engine's
    S.partNext[g] = g < c.G && c.partPeriod > 0 ? nowTick + c.partPeriod : 0;
  }
}

// Make for every peer of the engine at tick 0, one thread per lane.
__global__ void k_init(DevPtrs S, DevCfg c, int PP, long long nLanes) {
  long long gl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gl >= nLanes) return;
  init_lane(S, c, PP, gl, 0);
  const int g = (int)(gl / PP), p = (int)(gl % PP);
  if (g < c.G && p < c.P) {                    // logs[0], the entry every log begins with (raft.go:710-715)
    const Rows lg = rows_of(S.logRows, c, g);
    *lg.termp(0, p) = 0; *lg.cmdp(0, p) = 0;
  }
}

// A fresh set of peers (config.go:60-96: a new tester config) in each listed group at tick nowTick: one block per
// group; the lanes first, then the group's log rows, snapshot rows and tester record are cleared.
__global__ void k_reset_groups(DevPtrs S, DevCfg c, int PP, const int *groups, int nowTick) {
  const int g = groups[blockIdx.x];
  if ((int)threadIdx.x < PP) init_lane(S, c, PP, (long long)g * PP + threadIdx.x, nowTick);
  const size_t rowInts = group_log_bytes(c) / 4;
  int *lr = (int *)(S.logRows + (size_t)g * group_log_bytes(c));
  for (size_t i = threadIdx.x; i < rowInts; i += blockDim.x) lr[i] = 0;
  if (S.snapRows) {
    int *sr = (int *)(S.snapRows + (size_t)g * group_log_bytes(c));
    for (size_t i = threadIdx.x; i < rowInts; i += blockDim.x) sr[i] = 0;
  }
  for (int i = threadIdx.x; i < c.L; i += blockDim.x) S.applyRec[(size_t)g * c.L + i] = 0;
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

// Counters and safety summary of the engine's groups, packed for one reduction: out[0 .. RG_N_COUNTERS) = the counter
// block (group-ticks from the host's clock), out[RG_N_COUNTERS ..) = RG_S_* (zeroed by the caller).  One thread per group.
__global__ void k_summary(DevPtrs S, int G, int P, int PP, unsigned long long groupTicks, unsigned long long *out) {
  __shared__ unsigned long long acc[RG_N_SUMMARY];
  if (threadIdx.x < RG_N_SUMMARY) acc[threadIdx.x] = 0;
  __syncthreads();
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) {
    const unsigned f = S.gflags[g];
    unsigned long long rounds = 0;
    for (int p = 0; p < P; p++) rounds += (unsigned)S.roundSeq[(size_t)g * PP + p];
    const unsigned long long rec = (unsigned long long)S.recMax[g];
    atomicAdd(&acc[RG_S_ROUNDS_ALL], rounds);
    atomicAdd(&acc[RG_S_RECORDED_ALL], rec);
    if (f) {
      for (int b = 0; b < RG_N_FLAG_BITS; b++) if ((f >> b) & 1u) atomicAdd(&acc[RG_S_BIT_GROUPS + b], 1ull);
      atomicAdd(&acc[RG_S_FLAGGED_GROUPS], 1ull);
      atomicAdd(&acc[RG_S_ROUNDS_FLAGGED], rounds);
      atomicAdd(&acc[RG_S_RECORDED_FLAGGED], rec);
    }
  }
  __syncthreads();
  if (threadIdx.x < RG_N_SUMMARY && acc[threadIdx.x]) atomicAdd(&out[RG_N_COUNTERS + threadIdx.x], acc[threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x < RG_N_COUNTERS) out[threadIdx.x] = threadIdx.x == RG_C_GROUP_TICKS ? groupTicks : S.counters[threadIdx.x];
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
    v[RG_F_DEAD] = (f & F_DEAD) ? ((f & F_ZOMBIE) ? 2 : 1) : 0; v[RG_F_RESTART] = (f & F_RESTART) ? 1 : 0; v[RG_F_FIRST] = (f & F_FIRST) ? 1 : 0;
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

#endif
