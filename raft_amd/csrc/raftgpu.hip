// raftgpu.hip -- batched lockstep Raft engine for MI355X (gfx950): the C ABI of include/raftgpu.h (host side).
// Device code: rg_state.h (state layout, the reference's statements as lane functions), rg_tick.h (the tick
// kernel), rg_kernels.h (host verbs, batched RPC handlers, checks and digests).
//
// Mapping.  One lane per (group, peer).  In the tick kernel a group's P peers sit in P adjacent lanes of one
// 64-wide wavefront (12 five-peer groups per wave), so a wave never needs another wave: every cross-peer
// exchange is a bpermute or a ballot slice.  The tick kernel loads a peer's words into registers and LDS once,
// advances n ticks, and stores them back; logs and the leader matrices stay in HBM and are touched only by the
// events that need them.
//
// Log layout.  Entry i of ALL peers of a group shares one row {int32 term[P]; uint64 cmd[P]} (64 B for P = 5),
// rows consecutive in i.  A leader replicating entry i to its followers and the apply reads of index i
// therefore fall in the same cache line.  The persisted log shares these rows.
//
// Semantics.  A tick is: control (partition reshuffle, churn, client Starts), actors in a seeded cyclic peer
// order (election loop iteration, then leader loop), apply, one-leader check -- the reference's statements
// (src/raft/raft.go) under the serialisation fixed by the lockstep model.  The reference runs a round's RPCs
// one after the other and stops at the first reply with a higher term (raft.go:482-487, 622-627).  Here all
// targets of a round evaluate their handler at once without touching state, the break point is found with a
// ballot, and only targets up to it commit their effects: the result is the sequential one.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/raftgpu.h"

typedef unsigned long long u64;

#include "rg_state.h"
#include "rg_tick.h"
#include "rg_kernels.h"

// ---------------------------------------------------------------- fixed device scratch of rg_step_noalloc
// The relay queue and staging area of a relay launch (see k_tick) for callers that may not allocate.  One copy
// per GPU, held until the process ends.  The name below is a host-side stand-in, not a device address: host code
// uses the per-device address hipGetSymbolAddress returns (scratch_of), never the name.
#define RG_SCRATCH_INTS (RG_NOALLOC_SCRATCH_BYTES / 4u)
static __device__ __align__(256) int scratch[RG_SCRATCH_INTS];
#define RG_MAX_DEVICES 64
static int *g_scratchPtr[RG_MAX_DEVICES];               // per device: address of that device's copy of scratch
// Per device: recorded after every launch that uses the array; the next such launch, of whichever engine and on
// whichever stream, waits for it first, so launches that share the array never overlap on the GPU.  Created once per
// device (rg_create); waiting for and recording an event allocates nothing.
static hipEvent_t g_scratchDone[RG_MAX_DEVICES];
static std::mutex g_scratchMu;                          // also makes wait + launch + record one step per device
// Address of the current device's copy; looked up once per device, with that device current.
static int scratch_of(int device, int **out) {
  if (device < 0 || device >= RG_MAX_DEVICES) return RG_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_scratchMu);
  if (!g_scratchPtr[device]) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return RG_ERR_HIP;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return RG_ERR_HIP;
    g_scratchDone[device] = ev;
    g_scratchPtr[device] = (int *)p;
  }
  *out = g_scratchPtr[device];
  return RG_OK;
}

// ---------------------------------------------------------------- host side
// Launches of at most this many ticks run only the groups that are due (k_tick_due).  Measured on 65,536 five-peer groups
// of the benchmark workload, kernel time of a launch of 1 / 4 / 8 / 12 / 16 / 20 ticks: 103 / 142 / 166 / 180 / 188 / 203 us full,
// 42 / 76 / 117 / 150 / 183 / 212 us due (a sixth of the groups is due in one tick, nearly all of them in twelve).
#ifndef RG_DUE_MAX_TICKS
#define RG_DUE_MAX_TICKS 12
#endif
struct rg_engine {
  rg_config cfg;
  int G, P, L, PP, Gpad, gpb, rowBytes, cmdOff;
  bool cols;            // log layout: per-peer columns instead of rows (rg_state.h, Rows)
  size_t groupBytes;    // bytes of one group's log in either layout
  int slotWaves;        // wavefronts of the tick kernel the device holds at once
  int tickBlocks;       // grid of a plain launch of the tick kernel (DevCfg::nbA blocks of gwA groups per wave, then blocks of gwB)
  int relaySets, relayGw;   // sets of a relay launch and groups per wave in them; 0 when the groups fit the chip at once
  int *relay;           // device: [0] next item, [1 + set] chunks done
  int *relayStage;      // device: per-group words in transit between the blocks of a relay launch
  int *scratchDev;      // this device's copy of the fixed scratch array (rg_step_noalloc)
  long long nLanes;
  DevPtrs S;
  DevPtrs *dS;          // device copy of S: the tick kernel reads the cold pointers through it
  HotPtrs H;
  DevCfg dc;
  std::vector<void *> allocs;
  hipStream_t ownStream, stream;
  hipEvent_t ev0, ev1;
  struct TimedLaunch { hipEvent_t a, b; int weight; bool due; };   // weight: the launches this sample stands for
  std::vector<TimedLaunch> pendingEv;
  double stepMs;
  int dueUntimed, dueTimeEvery;      // due launches since the last timed one; one in so many is timed
  double lastDueMs;                  // the last timed due launch
  bool dueSampled;                   // a due launch has been timed
  long long launches;
  long long now;
  bool derivedDirty, syncDirty;
  // Due launches (k_tick_due): short launches run only the groups whose `wake` word falls inside the launch.
  bool lazy;            // some groups' stored words lag behind `now` (skipped by a due launch): settle() before reading them
  bool wakeDirty;       // the host wrote state behind the wake words' back: the next launch is a full one, which rewrites them all
  int dueWin;           // groups per block window of a due launch
  int dueMaxTicks;      // launches of at most this many ticks are due launches (0: never)
  std::vector<unsigned> drained;  // per (g,p): ApplyMsgs handed out by rg_drain_applied so far
  rg_apply_msg *applyRing; unsigned *applyHead; int applyCap;
  unsigned *netCfg;               // device: per-group net rates, allocated by the first rg_set_unreliable_group
  rg_sched_op *sched; int schedN; size_t schedCap; // device: the scenario schedule, and the room of its buffer
  std::vector<rg_sched_op> schedHost;
  std::mutex mu;
  // staging
  HostCmd *dCmd, *hCmd; size_t dCmdCap;     // device and pinned host staging of host commands
  StartIn *dIn, *hIn; StartOut *dOut, *hOut; size_t startCap;   // the same for rg_start_batch's compact records
  std::vector<unsigned> cmdStamp; std::vector<int> cmdSeen; unsigned cmdEpoch;   // per group: batch that last named it, and how often so far
  // the tick kernel compiled for this engine's configuration (rg_specialize), and what it was compiled for
  bool specWanted;
  hipModule_t specMod; hipFunction_t specFn, specDueFn;
  bool specExt, specLean; DevCfg specFor;
  // the cross-GPU exchange: counters + safety summary packed in one device buffer, reduced in place
  u64 *redBuf;                    // device: [RG_N_COUNTERS + RG_N_SUMMARY]
  void *comm;                     // ncclComm_t made by rg_comm_init, or null
};

// byte offset of entry i of peer p from the first byte of its group's log, in the engine's layout (rg_state.h, Rows)
static inline size_t host_toff(const rg_engine *e, int i, int p) {
  return e->cols ? ((size_t)p * e->L + i) * 4 : (size_t)i * e->rowBytes + 4 * (size_t)p;
}
static inline size_t host_coff(const rg_engine *e, int i, int p) {
  return (size_t)e->cmdOff + (e->cols ? ((size_t)p * e->L + i) * 8 : (size_t)i * e->rowBytes + 8 * (size_t)p);
}
// bytes from the group's first byte that hold its entries [0, n)
static inline size_t host_span(const rg_engine *e, int n) { return e->cols ? e->groupBytes : (size_t)n * e->rowBytes; }

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
  case RG_ERR_OVERRUN: return "apply ring overrun: the oldest ApplyMsgs were overwritten before they were drained";
  case RG_ERR_SCRATCH: return "engine too large for the fixed device scratch of rg_step_noalloc";
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
  if (c->log_layout < RG_LOG_AUTO || c->log_layout > RG_LOG_COLUMNS) return false;
  if (c->n_groups < 1 || c->n_peers < 2 || c->n_peers > RG_MAX_PEERS || c->log_cap < 2) return false;
  if (c->start_period < 0 || c->part_period < 0 || c->churn_period < 0) return false;
  if (c->start_period > 0 && (c->start_max < 0 || c->start_max + 1 > c->log_cap)) return false;
  if (c->apply_ring_cap < 0) return false;
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

// The network model's three rates: the only words of the device configuration that change after rg_create.
static void fill_net(rg_engine *e) {
  DevCfg &d = e->dc; const rg_config &c = e->cfg;
  d.reqDrop = c.req_drop_permille; d.replyDrop = c.reply_drop_permille; d.replyDelay = c.reply_delay_permille;
}

// Device configuration and launch layout (packing, relay sets), computed once in rg_create: the relay buffers are
// sized from it.
static void fill_devcfg(rg_engine *e) {
  DevCfg &d = e->dc; const rg_config &c = e->cfg;
  d.G = e->G; d.P = e->P; d.L = e->L; d.rowBytes = e->rowBytes; d.cmdOff = e->cmdOff;
  d.cols = e->cols ? 1 : 0; d.groupBytes = (int)e->groupBytes;
  d.groupBase = c.group_base; d.seedKey = sm64(c.seed);
  fill_net(e);
  d.startPeriod = c.start_period; d.startFirst = c.start_first_tick; d.startMax = c.start_max; d.startMode = c.start_mode;
  d.partPeriod = c.part_period; d.partJitter = c.part_jitter;
  d.churnPeriod = c.churn_period; d.churnDisc = c.churn_disconnect_permille; d.churnRestart = c.churn_restart_permille; d.churnCrash = c.churn_crash_permille;
  d.acctBytes = c.account_bytes != 0;
  // Packing of groups into waves.  Whole batches that fill every wave slot use full waves (64/P groups).  A
  // launch smaller than one batch is spread over all wave slots; what is left after whole batches is spread
  // over half of them (measured on the 65,536 x 5 workload: 2 blocks per CU finish the remainder earliest,
  // because it starts while the last whole batch is still draining).
  const int gwMax = 64 / e->P;
  const long long cap = (long long)e->slotWaves * gwMax;
  const int fullBatches = (int)(e->G / cap);
  d.gwA = gwMax; d.nbA = fullBatches * (e->slotWaves / 4);
  if (const char *ov = getenv("RAFTGPU_FULL_BLOCKS")) { int v = atoi(ov); if (v >= 0 && (long long)v * 4 * gwMax <= e->G) d.nbA = v; }   // tuning aid
  const int rest = e->G - d.nbA * 4 * gwMax;
  // A launch smaller than one batch on a reliable, unperturbed net: all groups move alike, so a wave's passes do not
  // grow with the number of its groups, and fewer, fuller waves (one per SIMD) run each pass faster (4,096 five-peer
  // groups: 12.2 ms with one group per wave, 9.75 ms with four).  With loss, partitions or churn the groups of a wave
  // differ and thin waves win (16,384 seven-peer groups under churn: 79.0 / 81.0 / 85.6 ms with 4 / 6 / 9 per wave).
  const bool alike = c.req_drop_permille == 0 && c.reply_drop_permille == 0 && c.reply_delay_permille == 0 && c.part_period == 0 && c.churn_period == 0;
  const int spread = d.nbA > 0 ? e->slotWaves / 2 : (alike ? std::max(4, e->slotWaves / 4) : e->slotWaves);
  int gwB = (rest + spread - 1) / spread;
  if (const char *ov = getenv("RAFTGPU_TAIL_GROUPS_PER_WAVE")) gwB = atoi(ov);   // tuning aid
  d.gwB = gwB < 1 ? 1 : (gwB > gwMax ? gwMax : gwB);
  e->tickBlocks = d.nbA + (rest + 4 * d.gwB - 1) / (4 * d.gwB);
  d.relaySets = 0; d.relayChunk = 0; d.lazy = 0;
  // Due launches: one block looks at a window of consecutive groups and runs those that are due, shared out among its waves.
  // A short launch costs a wave the same whether it holds one group or twelve, so the due groups should fill few waves: the
  // windows are as large as leaving one block per CU allows (65,536 groups: 256 per block, of which about 40 are due in a
  // one-tick launch of the benchmark workload), 256 at most.
  {
    const int cus = std::max(1, e->slotWaves / (4 * RG_TICK_WAVES));
    long long win = ((long long)e->G + cus - 1) / cus;
    if (const char *ov = getenv("RAFTGPU_DUE_WIN")) win = atoi(ov);           // tuning aid
    e->dueWin = (int)(win < 1 ? 1 : (win > 256 ? 256 : win));
    d.dueWin = e->dueWin;
    d.dueSplit = 2;                     // two blocks per window: the second runs the rounds beyond the first at the same time
    if (const char *ov = getenv("RAFTGPU_DUE_SPLIT")) { const int v = atoi(ov); if (v >= 1 && v <= 8) d.dueSplit = v; }   // tuning aid
    e->dueTimeEvery = 16;
    if (const char *ov = getenv("RAFTGPU_DUE_TIME_EVERY")) { const int v = atoi(ov); if (v >= 1) e->dueTimeEvery = v; }   // 1: every launch is timed
    e->dueMaxTicks = RG_DUE_MAX_TICKS;
    if (const char *ov = getenv("RAFTGPU_DUE_MAX_TICKS")) e->dueMaxTicks = atoi(ov);   // tuning / test aid; 0: every launch is a full one
    // (the lane offsets of a window must fit 32 bits)
    if ((unsigned long long)e->groupBytes * 256ull >= (1ull << 32) || (unsigned long long)e->L * 256ull * 8ull >= (1ull << 32)) e->dueMaxTicks = 0;
  }
  // More groups than the chip holds at once: long launches go through the relay queue (see k_tick).  There a set's
  // state passes from block to block within a launch, so no 128-byte line of any array may hold words of two
  // sets: the waves of a relay launch hold the most groups for which every array's bytes per set are whole lines.
  e->relaySets = 0; e->relayGw = 0;
  if (e->G > cap) {
    for (int gw = gwMax; gw >= gwMax - 1 && gw >= 1; gw--) {
      const long long gs = 4 * gw;                               // groups per set
      const bool whole = gs * e->PP * 4 % 128 == 0 && gs * e->PP * e->PP * 4 % 128 == 0 &&
                         gs * (long long)e->groupBytes % 128 == 0 && gs * e->L * 8 % 128 == 0 &&
                         gs * e->P * (long long)sizeof(rg_apply_msg) * e->applyCap % 128 == 0;
      if (whole) { e->relayGw = gw; e->relaySets = (int)((e->G + gs - 1) / gs); break; }
    }
  }
}

// Make for every peer: one kernel for the per-peer and per-group words, memsets for the rows.
// fresh: the arrays come straight from hipMalloc.  A reset leaves the log rows, snapshot rows and tester records as they
// are but for every log's entry 0: nothing reads a log beyond its length or a record beyond the group's highest recorded
// index before it has been written again (4.5 GB less to write for 65,536 five-peer groups with 1,024-entry logs: 0.8 ms).
static int init_state(rg_engine *e, bool fresh) {
  const long long nL = e->nLanes;
  DevPtrs &S = e->S;
  hipStream_t st = e->stream;
  if (fresh) {
    HIPCHK(hipMemsetAsync(S.logRows, 0, (size_t)e->G * e->groupBytes, st));
    if (S.snapRows) HIPCHK(hipMemsetAsync(S.snapRows, 0, (size_t)e->G * e->groupBytes, st));
    HIPCHK(hipMemsetAsync(S.applyRec, 0, (size_t)e->G * e->L * 8, st));
  }
  hipLaunchKernelGGL(k_init, dim3((unsigned)((nL + 255) / 256)), dim3(256), 0, st, S, e->dc, e->PP, nL);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(S.counters, 0, RG_CTR_SLOTS * 8, st));
  HIPCHK(hipMemsetAsync(S.flagSummary, 0, 8, st));
  if (e->applyCap > 0) HIPCHK(hipMemsetAsync(e->applyHead, 0, (size_t)e->nLanes * sizeof(unsigned), st));
  if (e->netCfg) HIPCHK(hipMemsetAsync(e->netCfg, 0, (size_t)e->Gpad * sizeof(unsigned), st));
  HIPCHK(hipMemsetAsync(S.schedPos, 0, (size_t)e->Gpad * sizeof(int), st));      // tick 0: the schedule starts over
  HIPCHK(hipStreamSynchronize(st));
  e->now = 0; e->derivedDirty = false; e->syncDirty = false;
  e->lazy = false; e->wakeDirty = false;           // k_init left every wake word at 0: all groups are looked at
  std::fill(e->drained.begin(), e->drained.end(), 0);
  return RG_OK;
}

static void comm_release(void *comm);   // ncclCommDestroy, below with the rest of the RCCL plumbing
static void free_engine(rg_engine *e) {
  for (auto &pr : e->pendingEv) { hipEventDestroy(pr.a); hipEventDestroy(pr.b); }
  for (void *p : e->allocs) hipFree(p);
  if (e->dCmd) hipFree(e->dCmd);
  if (e->hCmd) hipHostFree(e->hCmd);
  if (e->hIn) hipHostFree(e->hIn);
  if (e->hOut) hipHostFree(e->hOut);
  if (e->ownStream) hipStreamDestroy(e->ownStream);
  if (e->specMod) hipModuleUnload(e->specMod);
  if (e->comm) comm_release(e->comm);
  delete e;
}

// The separate snapshot rows exist only once something needs them: an imported Persister blob or a
// host write to the log arrays.
static int ensure_snap_rows(rg_engine *e) {
  if (e->S.snapRows) return RG_OK;
  const size_t n = (size_t)e->G * e->groupBytes;
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
  e->Gpad = (e->G + e->gpb - 1) / e->gpb * e->gpb + e->gpb;   // thinner waves reach up to one block past the last group
  e->slotWaves = prop.multiProcessorCount * 4 * RG_TICK_WAVES;
  if (const char *ov = getenv("RAFTGPU_SLOT_WAVES")) { int v = atoi(ov); if (v >= 4) e->slotWaves = v / 4 * 4; }   // test aid: pretend the chip is smaller
  e->nLanes = (long long)e->Gpad * e->PP;
  e->cmdOff = (e->P * 4 + 7) & ~7;
  e->rowBytes = (e->cmdOff + e->P * 8 + 15) & ~15;
  e->groupBytes = (size_t)e->L * e->rowBytes;
  // Log layout (rg_state.h, Rows).  Rows serve short logs under steady replication; logs that are replayed whole
  // (restarts of the crash / restart churn, raft.go:717 with :672-688) or walked far (long logs) are per-peer columns.
  e->cols = cfg->log_layout == RG_LOG_COLUMNS || (cfg->log_layout == RG_LOG_AUTO && (cfg->log_cap >= 2048 || cfg->churn_period > 0));
  if (e->cols) {
    e->cmdOff = (e->P * e->L * 4 + 15) & ~15;               // the commands follow the P term arrays
    e->groupBytes = ((size_t)e->cmdOff + (size_t)e->P * e->L * 8 + 15) & ~(size_t)15;
  }
  e->ownStream = nullptr; e->dCmd = nullptr; e->hCmd = nullptr; e->dCmdCap = 0; e->cmdEpoch = 0; e->stepMs = 0; e->launches = 0; e->dueUntimed = 0; e->lastDueMs = 0; e->dueSampled = false;
  if (hipStreamCreateWithFlags(&e->ownStream, hipStreamNonBlocking) != hipSuccess) { delete e; return RG_ERR_HIP; }
  e->stream = e->ownStream;
  DevPtrs &S = e->S;
  int rc = scratch_of(cfg->device, &e->scratchDev);
  if (rc != RG_OK) { hipStreamDestroy(e->ownStream); delete e; return rc; }
  int **i32s[] = {&S.term, &S.vote, &S.flags, &S.commit, &S.applied, &S.logLen, &S.timer, &S.timeout, &S.hbTimer, &S.elNext, &S.monNext,
                  &S.rngCtr, &S.roundSeq, &S.snapTerm, &S.snapVote, &S.snapLen, &S.snapSync, &S.lastTerm, &S.firstOwn, &S.commitTerm};
  for (int **p : i32s) if (rc == RG_OK) rc = dalloc(e, p, (size_t)e->nLanes);
  if (rc == RG_OK) rc = dalloc(e, &S.applyHash, (size_t)e->nLanes);
  if (rc == RG_OK) rc = dalloc(e, &S.nextIdx, (size_t)e->Gpad * e->PP * e->PP);
  if (rc == RG_OK) rc = dalloc(e, &S.matchIdx, (size_t)e->Gpad * e->PP * e->PP);
  if (rc == RG_OK) rc = dalloc(e, &S.logRows, (size_t)e->G * e->groupBytes);
  S.snapRows = nullptr;     // allocated when a blob is first imported (ensure_snap_rows)
  if (rc == RG_OK) rc = dalloc(e, &S.applyRec, (size_t)e->G * e->L);
  if (rc == RG_OK) rc = dalloc(e, &S.startSeq, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.partNext, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.partEpoch, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.recMax, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.schedPos, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.schedRegs, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.gflags, (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.wakeSettled, 2 * (size_t)e->Gpad);
  if (rc == RG_OK) rc = dalloc(e, &S.counters, (size_t)RG_CTR_SLOTS);
  if (rc == RG_OK) rc = dalloc(e, &S.flagSummary, (size_t)2);
  if (rc == RG_OK) rc = dalloc(e, &e->redBuf, (size_t)(RG_N_COUNTERS + RG_N_SUMMARY));
  if (rc != RG_OK) { free_engine(e); return rc; }
  e->drained.assign((size_t)e->G * e->P, 0);
  e->applyCap = cfg->apply_ring_cap; e->applyRing = nullptr; e->applyHead = nullptr; e->netCfg = nullptr; e->sched = nullptr; e->schedN = 0; e->schedCap = 0;
  if (e->applyCap > 0) {
    rc = dalloc(e, &e->applyRing, (size_t)e->G * e->P * e->applyCap);
    if (rc == RG_OK) rc = dalloc(e, &e->applyHead, (size_t)e->nLanes);
    if (rc != RG_OK) { free_engine(e); return rc; }
  }
  fill_devcfg(e);
  if (e->relaySets > 0) {
    rc = dalloc(e, &e->relay, (size_t)e->relaySets + 1);
    if (rc == RG_OK) rc = dalloc(e, &e->relayStage, (size_t)e->relaySets * 7 * ((e->gpb + 31) / 32 * 32));
    if (rc != RG_OK) { free_engine(e); return rc; }
  }
  e->H.relay = e->relay; e->H.relayStage = e->relayStage;
  e->H.applyRing = e->applyRing; e->H.applyHead = e->applyHead; e->H.applyCap = e->applyCap; e->H.netCfg = nullptr; e->H.sched = nullptr; e->H.schedN = 0;
  e->H.logRows = S.logRows; e->H.snapRows = S.snapRows; e->H.nextIdx = S.nextIdx; e->H.matchIdx = S.matchIdx; e->H.applyRec = S.applyRec;
  if (dalloc(e, &e->dS, 1) != RG_OK || hipMemcpy(e->dS, &e->S, sizeof(DevPtrs), hipMemcpyHostToDevice) != hipSuccess) { free_engine(e); return RG_ERR_NOMEM; }
  rc = init_state(e, true);
  if (rc != RG_OK) { free_engine(e); return rc; }
  { const char *v = getenv("RAFTGPU_SPECIALIZE"); e->specWanted = v && atoi(v) > 0; }   // as rg_specialize, compiled at the first launch
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
    if (hipEventSynchronize(pr.b) != hipSuccess) return RG_ERR_HIP;
    if (hipEventElapsedTime(&ms, pr.a, pr.b) != hipSuccess) return RG_ERR_HIP;
    e->stepMs += (double)ms * pr.weight;
    if (pr.due) e->lastDueMs = ms;
    hipEventDestroy(pr.a); hipEventDestroy(pr.b);
  }
  e->pendingEv.clear();
  return RG_OK;
}

extern "C" int rg_reset(rg_engine *e) {
  LOCK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  int rc = collect_events(e);
  if (rc != RG_OK) return rc;
  return init_state(e, false);
}

extern "C" int rg_set_stream(rg_engine *e, void *s) {
  LOCK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  e->stream = s ? (hipStream_t)s : e->ownStream;
  return RG_OK;
}

static int run_derive(rg_engine *e) {
  if (!e->derivedDirty) return RG_OK;
  e->wakeDirty = true;                       // k_derive marks every leader's commit rule for evaluation
  int blocks = (int)((e->nLanes + 255) / 256);
  hipLaunchKernelGGL(k_derive, dim3(blocks), dim3(256), 0, e->stream, e->S, e->dc, e->PP, e->syncDirty ? 1 : 0);
  HIPCHK(hipGetLastError());
  e->derivedDirty = false; e->syncDirty = false;
  return RG_OK;
}

// The per-wave counters of a launch are 32-bit words in LDS; the fastest of them (wire bytes) can grow by a few
// thousand per tick, so one launch covers at most this many ticks.
#define RG_MAX_LAUNCH_TICKS 500000
// Times are int32 milliseconds on the device (now = tick x RG_TICK_MS, plus timeouts and waits of up to a few
// seconds): the clock stops here, two simulated hours short of the overflow.
#define RG_MAX_NOW_TICKS ((long long)(0x7fffffff / RG_TICK_MS) - 720000)

// Ticks per chunk of a relay launch: chunks of at least RG_RELAY_MIN_TICKS, at most RG_RELAY_MAX_CHUNKS of them
// (a chunk boundary costs a store and a load of the state, a cold start of the LDS rings and an L2 write-back;
// measured on 65,536 five-peer groups x 20,000 ticks: 67.2 ms plain, 53.6 / 53.4 / 53.5 ms with 6 / 8 / 10 chunks,
// 59.7 ms with 24 (a later, faster kernel: 48.3 / 48.0 / 47.8 / 47.7 / 47.8 ms with 6 / 8 / 10 / 12 / 16); on 131,072 three-peer groups x 4,000 ticks at 1 % loss: 9.1 ms plain, 8.4 / 8.2 / 8.1 ms
// with 2 / 4 / 6 chunks).  0: plain launch.
#define RG_RELAY_MIN_TICKS 1000
#define RG_RELAY_MAX_CHUNKS 12
static int relay_chunk_ticks(const rg_engine *e, int n_ticks) {
  if (e->relaySets == 0 || e->schedN > 0) return 0;   // the schedule's per-group words do not travel through the relay's staging area
  int k = n_ticks / RG_RELAY_MIN_TICKS;
  if (k > RG_RELAY_MAX_CHUNKS) k = RG_RELAY_MAX_CHUNKS;
  if (const char *ov = getenv("RAFTGPU_RELAY_CHUNKS")) k = atoi(ov);   // tuning aid; 0 or 1: plain launches
  if (k < 2) return 0;
  return (n_ticks + k - 1) / k;
}

// ---------------------------------------------------------------- a tick kernel compiled for one engine
// rg_specialize: the device sources (this library's own, embedded below as they were when it was built) are compiled
// again at run time through hiprtc with everything rg_create fixed as compile-time constants (RG_SPEC_* in
// rg_tick.h).  hiprtc is loaded when first asked for, so the library does not depend on it otherwise.
#if !defined(__HIP_DEVICE_COMPILE__)
__asm__(".pushsection .rodata\n"
        ".hidden rg_src_abi\n.global rg_src_abi\nrg_src_abi:\n.incbin \"raftgpu.h\"\n.byte 0\n"
        ".hidden rg_src_state\n.global rg_src_state\nrg_src_state:\n.incbin \"rg_state.h\"\n.byte 0\n"
        ".hidden rg_src_tick\n.global rg_src_tick\nrg_src_tick:\n.incbin \"rg_tick.h\"\n.byte 0\n"
        ".popsection\n");
#endif
extern "C" const char rg_src_abi[], rg_src_state[], rg_src_tick[];
static const char kSpecUnit[] =
    "#include \"raftgpu.h\"\ntypedef unsigned long long u64;\n#include \"rg_state.h\"\n#include \"rg_tick.h\"\n";
// what hiprtc's built-in headers leave to the standard ones raftgpu.h includes
static const char kSpecStdint[] =
    "#pragma once\ntypedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
    "typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;\n";
static const char kSpecStddef[] = "#pragma once\n";

struct Rtc {
  void *lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) logSize = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) codeSize = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
};
static const Rtc *rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define RTC_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.lib, #sym)
    RTC_SYM(create, hiprtcCreateProgram); RTC_SYM(compile, hiprtcCompileProgram); RTC_SYM(logSize, hiprtcGetProgramLogSize);
    RTC_SYM(log, hiprtcGetProgramLog); RTC_SYM(codeSize, hiprtcGetCodeSize); RTC_SYM(code, hiprtcGetCode); RTC_SYM(destroy, hiprtcDestroyProgram);
#undef RTC_SYM
    if (!(r.create && r.compile && r.logSize && r.log && r.codeSize && r.code && r.destroy)) r.lib = nullptr;
  });
  return r.lib ? &r : nullptr;
}

// the fields of DevCfg a specialised kernel has as constants
static bool spec_same(const DevCfg &a, const DevCfg &b) {
  return a.P == b.P && a.L == b.L && a.rowBytes == b.rowBytes && a.cmdOff == b.cmdOff && a.cols == b.cols && a.groupBytes == b.groupBytes && a.reqDrop == b.reqDrop && a.replyDrop == b.replyDrop &&
         a.replyDelay == b.replyDelay && a.startPeriod == b.startPeriod && a.startFirst == b.startFirst && a.startMax == b.startMax &&
         a.startMode == b.startMode && a.partPeriod == b.partPeriod && a.partJitter == b.partJitter && a.churnPeriod == b.churnPeriod &&
         a.churnDisc == b.churnDisc && a.churnRestart == b.churnRestart && a.churnCrash == b.churnCrash && a.acctBytes == b.acctBytes;
}

// hiprtc: the embedded sources with these options -> a code object.  Code objects already compiled in this process are
// kept by option string, so engines of one configuration share the compile.
static int compile_spec(const Rtc *r, const std::vector<std::string> &opts, std::vector<char> &code) {
  static std::mutex cacheMu;
  static std::unordered_map<std::string, std::vector<char>> cache;
  std::string key;
  for (auto &o : opts) { key += o; key += ' '; }
  {
    std::lock_guard<std::mutex> lk(cacheMu);
    auto it = cache.find(key);
    if (it != cache.end()) { code = it->second; return RG_OK; }
  }
  std::vector<const char *> optv;
  for (auto &o : opts) optv.push_back(o.c_str());
  const char *hdrs[] = {rg_src_abi, rg_src_state, rg_src_tick, kSpecStdint, kSpecStddef};
  const char *names[] = {"raftgpu.h", "rg_state.h", "rg_tick.h", "stdint.h", "stddef.h"};
  hiprtcProgram prog;
  if (r->create(&prog, kSpecUnit, "raftgpu_spec.hip", 5, hdrs, names) != HIPRTC_SUCCESS) return RG_ERR_HIP;
  if (r->compile(prog, (int)optv.size(), optv.data()) != HIPRTC_SUCCESS) {
    size_t n = 0;
    if (r->logSize(prog, &n) == HIPRTC_SUCCESS && n > 1) { std::string log(n, 0); r->log(prog, &log[0]); fprintf(stderr, "raftgpu: rg_specialize: hiprtc: %s\n", log.c_str()); }
    r->destroy(&prog);
    return RG_ERR_HIP;
  }
  size_t n = 0;
  if (r->codeSize(prog, &n) == HIPRTC_SUCCESS && n > 0) { code.resize(n); if (r->code(prog, code.data()) != HIPRTC_SUCCESS) code.clear(); }
  r->destroy(&prog);
  if (code.empty()) return RG_ERR_HIP;
  std::lock_guard<std::mutex> lk(cacheMu);
  if (cache.size() >= 256) cache.clear();
  cache[key] = code;
  return RG_OK;
}

// RAFTGPU_SPEC_DEFS reaches a compiler whose option parser ends the PROCESS on an option it does not know, and this
// library never aborts: only the options the tuning sweeps use get through.  -DRG_<NAME>[=<integer>], -O0..-O3 / -Os,
// -f[no-]unroll-loops, and `-mllvm <opt>` for the back-end switches listed here with the values listed here.
static bool spec_value_in(const std::string &v, std::initializer_list<const char *> ok) {
  for (const char *o : ok) if (v == o) return true;
  return false;
}
static bool spec_all_digits(const std::string &v) {
  if (v.empty() || v.size() > 9) return false;
  for (char ch : v) if (ch < '0' || ch > '9') return false;
  return true;
}
static bool spec_mllvm_ok(const std::string &t) {
  if (t.empty() || t[0] != '-') return false;
  const size_t eq = t.find('=');
  const std::string name = t.substr(1, eq == std::string::npos ? std::string::npos : eq - 1);
  const std::string val = eq == std::string::npos ? "" : t.substr(eq + 1);
  if (name == "amdgpu-sched-strategy") return spec_value_in(val, {"max-ilp", "max-memory-clause"});
  if (name == "amdgpu-schedule-relaxed-occupancy" || name == "amdgpu-early-inline-all") return spec_value_in(val, {"true", "false"});
  if (name == "amdgpu-sdwa-peephole" || name == "amdgpu-spill-sgpr-to-vgpr") return spec_value_in(val, {"0", "1"});
  if (name == "tail-dup-size") return spec_all_digits(val);
  if (name == "split-spill-mode") return spec_value_in(val, {"default", "size", "speed"});
  if (name == "disable-phi-elim-edge-splitting") return eq == std::string::npos;
  return false;
}
static bool spec_defs_ok(const std::vector<std::string> &toks) {
  for (size_t i = 0; i < toks.size(); i++) {
    const std::string &t = toks[i];
    if (t == "-mllvm") { if (i + 1 >= toks.size() || !spec_mllvm_ok(toks[i + 1])) return false; i++; continue; }
    if (spec_value_in(t, {"-O0", "-O1", "-O2", "-O3", "-Os", "-fno-unroll-loops", "-funroll-loops"})) continue;
    if (t.compare(0, 5, "-DRG_") == 0) {
      size_t j = 5;
      while (j < t.size() && ((t[j] >= 'A' && t[j] <= 'Z') || (t[j] >= '0' && t[j] <= '9') || t[j] == '_')) j++;
      if (j == 5) return false;
      if (j == t.size()) continue;
      if (t[j] != '=') return false;
      std::string v = t.substr(j + 1);
      if (!v.empty() && v[0] == '-') v = v.substr(1);
      if (!spec_all_digits(v)) return false;
      continue;
    }
    return false;
  }
  return true;
}

// Compiles and loads k_tick_spec for configuration c of the engine (device of the engine current).
static int build_spec(rg_engine *e, bool ext, bool lean, const DevCfg &c) {
  const Rtc *r = rtc();
  if (!r) return RG_ERR_STATE;
  if (e->specMod) {                                          // launches of the old kernel may still be running
    HIPCHK(hipStreamSynchronize(e->stream));
    hipModuleUnload(e->specMod);
    e->specMod = nullptr; e->specFn = nullptr; e->specDueFn = nullptr;
  }
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, e->cfg.device));
  std::string arch = prop.gcnArchName;                       // "gfx950:sramecc+:xnack-"
  arch = "--offload-arch=" + arch.substr(0, arch.find(':'));
  std::vector<std::string> opts = {arch, "-O3", "-std=c++17", "-Wno-unused-value"};
  auto def = [&](const char *name, long long v) { opts.push_back(std::string("-DRG_SPEC_") + name + "=" + std::to_string(v)); };
  def("P", c.P); def("EXT", ext); def("LEAN", lean); def("L", c.L); def("ROWBYTES", c.rowBytes); def("CMDOFF", c.cmdOff); def("COLS", c.cols); def("GROUPBYTES", c.groupBytes);
  def("REQDROP", c.reqDrop); def("REPLYDROP", c.replyDrop); def("REPLYDELAY", c.replyDelay);
  def("STARTPERIOD", c.startPeriod); def("STARTFIRST", c.startFirst); def("STARTMAX", c.startMax); def("STARTMODE", c.startMode);
  def("PARTPERIOD", c.partPeriod); def("PARTJITTER", c.partJitter);
  def("CHURNPERIOD", c.churnPeriod); def("CHURNDISC", c.churnDisc); def("CHURNRESTART", c.churnRestart); def("CHURNCRASH", c.churnCrash);
  def("ACCTBYTES", c.acctBytes);
  if (const char *x = getenv("RAFTGPU_SPEC_DEFS")) {          // tuning aid: extra compiler options, blank-separated
    const std::string t = x;
    std::vector<std::string> extra;
    for (size_t i = 0; i < t.size();) {
      size_t j = t.find(' ', i);
      if (j == std::string::npos) j = t.size();
      if (j > i) extra.push_back(t.substr(i, j - i));
      i = j + 1;
    }
    if (!spec_defs_ok(extra)) {
      fprintf(stderr, "raftgpu: rg_specialize: RAFTGPU_SPEC_DEFS=\"%s\" holds an option that is not on the list of tuning options; nothing was compiled\n", x);
      return RG_ERR_ARG;
    }
    opts.insert(opts.end(), extra.begin(), extra.end());
  }
  std::vector<char> code;
  int rc = compile_spec(r, opts, code);
  if (rc != RG_OK) return rc;
  HIPCHK(hipModuleLoadData(&e->specMod, code.data()));
  if (hipModuleGetFunction(&e->specFn, e->specMod, "k_tick_spec") != hipSuccess ||
      hipModuleGetFunction(&e->specDueFn, e->specMod, "k_tick_spec_due") != hipSuccess) { hipModuleUnload(e->specMod); e->specMod = nullptr; e->specFn = nullptr; e->specDueFn = nullptr; return RG_ERR_HIP; }
  e->specExt = ext; e->specLean = lean; e->specFor = c;
  return RG_OK;
}

// One launch of the tick kernel.  relay / relayStage: the queue and staging area a relay launch uses.
// mayCompile: a specialised kernel that no longer fits the configuration may be rebuilt here (never on the path that
// must not allocate).
static int launch_kernel(rg_engine *e, int n_ticks, int *relay, int *relayStage, bool mayCompile, bool forceFull = false) {
  DevCfg dc = e->dc;
  HotPtrs H = e->H;
  H.relay = relay; H.relayStage = relayStage;
  int blocks = e->tickBlocks;
  const DevPtrs *dS = e->dS;
  if (!forceFull && !e->wakeDirty && n_ticks > 0 && n_ticks <= e->dueMaxTicks) {
    // a due launch: the groups with an event inside the launch, found from their wake words by the kernel itself
    const unsigned dblocks = (unsigned)(((long long)e->G + e->dueWin - 1) / e->dueWin) * (unsigned)dc.dueSplit;
    dc.lazy = e->lazy ? 1 : 0;
    const bool extD = dc.acctBytes || H.applyCap > 0 || H.netCfg || H.schedN > 0;
    if (getenv("RAFTGPU_TRACE_LAUNCH")) fprintf(stderr, "raftgpu: k_tick_due ticks=%d blocks=%u window=%d lazy=%d\n", n_ticks, dblocks, e->dueWin, dc.lazy);
    if (e->specWanted) {                 // the kernel compiled for this configuration, if there is one that fits (never compiled here)
      const bool leanD = dc.churnPeriod == 0 && dc.startMode == RG_START_ONE && !extD;
      if (e->specDueFn && e->specExt == extD && e->specLean == leanD && spec_same(e->specFor, dc)) {
        int t0 = (int)e->now, n = n_ticks;
        void *params[] = {(void *)&e->S, (void *)&H, (void *)&dc, (void *)&t0, (void *)&n};
        HIPCHK(hipModuleLaunchKernel(e->specDueFn, dblocks, 1, 1, 256, 1, 1, 0, e->stream, params, nullptr));
        e->lazy = true;
        return RG_OK;
      }
    }
#define DUE_LAUNCH2(PEERS, COLS) do { \
    if (extD) hipLaunchKernelGGL((k_tick_due<PEERS, true, COLS>), dim3(dblocks), dim3(256), 0, e->stream, e->S, H, dc, (int)e->now, n_ticks); \
    else hipLaunchKernelGGL((k_tick_due<PEERS, false, COLS>), dim3(dblocks), dim3(256), 0, e->stream, e->S, H, dc, (int)e->now, n_ticks); } while (0)
#define DUE_LAUNCH(PEERS) do { if (e->cols) DUE_LAUNCH2(PEERS, true); else DUE_LAUNCH2(PEERS, false); } while (0)
#ifdef RG_ONLY_P                                   // diagnostic builds: one peer count, a tenth of the compile time
    if (e->P != RG_ONLY_P) return RG_ERR_ARG;
    DUE_LAUNCH(RG_ONLY_P);
#else
    switch (e->P) {
    case 2: DUE_LAUNCH(2); break;
    case 3: DUE_LAUNCH(3); break;
    case 4: DUE_LAUNCH(4); break;
    case 5: DUE_LAUNCH(5); break;
    case 6: DUE_LAUNCH(6); break;
    case 7: DUE_LAUNCH(7); break;
    default: DUE_LAUNCH(8); break;
    }
#endif
#undef DUE_LAUNCH2
#undef DUE_LAUNCH
    HIPCHK(hipGetLastError());
    e->lazy = true;
    return RG_OK;
  }
  // a full launch brings every group up to its end and rewrites every wake word
  dc.lazy = e->lazy ? 1 : 0;
  struct Full { rg_engine *e; ~Full() { e->lazy = false; e->wakeDirty = false; } } fullDone{e};
  const int chunk = relay_chunk_ticks(e, n_ticks);
  if (chunk > 0) {
    dc.relaySets = e->relaySets; dc.relayChunk = chunk;
    dc.nbA = e->relaySets; dc.gwA = e->relayGw;
    blocks = e->relaySets * ((n_ticks + chunk - 1) / chunk);   // one block per item
    HIPCHK(hipMemsetAsync(relay, 0, (size_t)(e->relaySets + 1) * sizeof(int), e->stream));
  }
  if (getenv("RAFTGPU_TRACE_LAUNCH")) fprintf(stderr, "raftgpu: k_tick ticks=%d blocks=%d relay_sets=%d relay_chunk=%d groups_per_wave=%d\n", n_ticks, blocks, dc.relaySets, dc.relayChunk, dc.gwA);
  const bool lean = dc.churnPeriod == 0 && dc.startMode == RG_START_ONE;     // nothing the lean instantiation leaves out is in use
  const bool ext = dc.acctBytes || H.applyCap > 0 || H.netCfg || H.schedN > 0;
  if (e->specWanted) {
    const bool fits = e->specFn && e->specExt == ext && e->specLean == (lean && !ext) && spec_same(e->specFor, dc);
    if (!fits && mayCompile && build_spec(e, ext, lean && !ext, dc) != RG_OK) e->specWanted = false;   // said once by build_spec; the generic kernel serves
    if (e->specFn && e->specExt == ext && e->specLean == (lean && !ext) && spec_same(e->specFor, dc)) {
      int t0 = (int)e->now, n = n_ticks;
      void *params[] = {(void *)&dS, (void *)&H, (void *)&dc, (void *)&t0, (void *)&n};
      HIPCHK(hipModuleLaunchKernel(e->specFn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, e->stream, params, nullptr));
      return RG_OK;
    }
  }
#define TICK_LAUNCH2(PEERS, COLS) do { \
    if (ext) hipLaunchKernelGGL((k_tick<PEERS, true, false, COLS>), dim3(blocks), dim3(256), 0, e->stream, dS, H, dc, (int)e->now, n_ticks); \
    else if (lean) hipLaunchKernelGGL((k_tick<PEERS, false, true, COLS>), dim3(blocks), dim3(256), 0, e->stream, dS, H, dc, (int)e->now, n_ticks); \
    else hipLaunchKernelGGL((k_tick<PEERS, false, false, COLS>), dim3(blocks), dim3(256), 0, e->stream, dS, H, dc, (int)e->now, n_ticks); } while (0)
#define TICK_LAUNCH(PEERS) do { if (e->cols) TICK_LAUNCH2(PEERS, true); else TICK_LAUNCH2(PEERS, false); } while (0)
#ifdef RG_ONLY_P
  if (e->P != RG_ONLY_P) return RG_ERR_ARG;
  TICK_LAUNCH(RG_ONLY_P);
#else
  switch (e->P) {
  case 2: TICK_LAUNCH(2); break;
  case 3: TICK_LAUNCH(3); break;
  case 4: TICK_LAUNCH(4); break;
  case 5: TICK_LAUNCH(5); break;
  case 6: TICK_LAUNCH(6); break;
  case 7: TICK_LAUNCH(7); break;
  default: TICK_LAUNCH(8); break;
  }
#endif
#undef TICK_LAUNCH2
#undef TICK_LAUNCH
  HIPCHK(hipGetLastError());
  return RG_OK;
}

// After due launches the stored words of the groups they skipped lag behind `now`: their election-loop and monitor
// wake-up chains, and the client Starts of an engine with a device client, wait for the group's next span.  A full
// launch of zero ticks carries all of that out (the end of every span does), so that what the host then reads is what
// tick-by-tick execution leaves.
static int settle(rg_engine *e) {
  if (!e->lazy) return RG_OK;
  int rc = run_derive(e);
  if (rc != RG_OK) return rc;
  return launch_kernel(e, 0, e->relay, e->relayStage, false, true);
}
// The same where only pending client Starts matter (they write logs, persisted headers, counters and the overflow flag; the
// wake-up chains are times nobody but the tick kernel looks at).
static int settle_client(rg_engine *e) { return e->dc.startPeriod > 0 ? settle(e) : RG_OK; }

static int launch_ticks(rg_engine *e, int n_ticks) {
  // The two event packets around a launch cost the stream about 8 us, a fifth of a one-tick due launch.  Of the due
  // launches the first and then one in dueTimeEvery are timed; a timed one stands for the untimed ones before it.
  const bool due = !e->wakeDirty && n_ticks <= e->dueMaxTicks;          // what launch_kernel will decide
  if (due && e->dueSampled && e->dueUntimed + 1 < e->dueTimeEvery) {
    int rc = launch_kernel(e, n_ticks, e->relay, e->relayStage, true);
    if (rc != RG_OK) return rc;
    e->dueUntimed++;
    e->launches++;
    e->now += n_ticks;
    return RG_OK;
  }
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  if (hipEventCreate(&b) != hipSuccess) { hipEventDestroy(a); return RG_ERR_HIP; }
  // the pair is owned by pendingEv from here on (collect_events and free_engine destroy it), whatever happens below
  rg_engine::TimedLaunch tl; tl.a = a; tl.b = b; tl.due = due; tl.weight = due ? e->dueUntimed + 1 : 1;
  e->pendingEv.push_back(tl);
  if (due) { e->dueUntimed = 0; e->dueSampled = true; }
  HIPCHK(hipEventRecord(a, e->stream));
  int rc = launch_kernel(e, n_ticks, e->relay, e->relayStage, true);
  if (rc != RG_OK) return rc;
  HIPCHK(hipEventRecord(b, e->stream));
  e->launches++;
  e->now += n_ticks;
  // group-ticks are counted on the host side of the counter block
  return RG_OK;
}

extern "C" int rg_specialize(rg_engine *e) {
  LOCK(e);
  const bool lean = e->dc.churnPeriod == 0 && e->dc.startMode == RG_START_ONE;
  const bool ext = e->dc.acctBytes || e->H.applyCap > 0 || e->H.netCfg || e->H.schedN > 0;
  int rc = build_spec(e, ext, lean && !ext, e->dc);
  if (rc == RG_OK) e->specWanted = true;
  return rc;
}

extern "C" int rg_step(rg_engine *e, int32_t n_ticks) {
  LOCK(e);
  if (n_ticks < 0) return RG_ERR_ARG;
  if (n_ticks == 0) return RG_OK;
  if (e->now + n_ticks > RG_MAX_NOW_TICKS) return RG_ERR_RANGE;
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

// Words of the relay queue (next item + one per set) rounded up to whole 128-byte lines, and of the staging area.
static size_t relay_queue_ints(const rg_engine *e) { return ((size_t)e->relaySets + 1 + 31) / 32 * 32; }
static size_t relay_stage_ints(const rg_engine *e) { return (size_t)e->relaySets * 7 * ((e->gpb + 31) / 32 * 32); }

// rg_step for callers that may not allocate: the same kernels, with the relay queue and staging area in the fixed
// device array instead of the buffers rg_create allocated, no events (the launch is not timed), no host or device
// allocation of any kind.
extern "C" int rg_step_noalloc(rg_engine *e, int32_t n_ticks) {
  LOCK(e);
  if (n_ticks < 0) return RG_ERR_ARG;
  if (n_ticks == 0) return RG_OK;
  if (e->now + n_ticks > RG_MAX_NOW_TICKS) return RG_ERR_RANGE;
  bool relays = false;                                   // does any launch of this call go through the relay queue?
  for (int rem = n_ticks; rem > 0; rem -= RG_MAX_LAUNCH_TICKS)
    if (relay_chunk_ticks(e, rem < RG_MAX_LAUNCH_TICKS ? rem : RG_MAX_LAUNCH_TICKS) > 0) relays = true;
  if (relays && relay_queue_ints(e) + relay_stage_ints(e) > (size_t)RG_SCRATCH_INTS) return RG_ERR_SCRATCH;
  int rc = run_derive(e);
  if (rc != RG_OK) return rc;
  int *const queue = e->scratchDev, *const stage = e->scratchDev + relay_queue_ints(e);
  while (n_ticks > 0) {
    const int n = n_ticks < RG_MAX_LAUNCH_TICKS ? n_ticks : RG_MAX_LAUNCH_TICKS;
    if (relay_chunk_ticks(e, n) > 0) {
      // The array is the GPU's, not the engine's: this launch starts after the last one that used it has ended,
      // whichever engine and stream that was.
      std::lock_guard<std::mutex> lk(g_scratchMu);
      hipEvent_t done = g_scratchDone[e->cfg.device];
      HIPCHK(hipStreamWaitEvent(e->stream, done, 0));
      rc = launch_kernel(e, n, queue, stage, false);    // clears the queue words it uses; the staging area is written before it is read
      if (rc != RG_OK) return rc;
      HIPCHK(hipEventRecord(done, e->stream));
    } else {
      rc = launch_kernel(e, n, queue, stage, false);    // a plain launch: no scratch
      if (rc != RG_OK) return rc;
    }
    e->now += n;
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
  // (due launches no sample stands for yet: at the time of the last one that was timed)
  if (total_ms) *total_ms = e->stepMs + e->lastDueMs * e->dueUntimed;
  if (launches) *launches = e->launches;
  return RG_OK;
}

// Level of each of n requests = how many earlier requests of the batch name its group (groups[i * stride]); returns the
// highest.  Per-group stamps of the batch instead of a map or a clear: O(n).
static int batch_levels(rg_engine *e, const int32_t *groups, size_t stride, size_t n, std::vector<int> &level) {
  level.assign(n, 0);
  int maxLevel = 0;
  if (n < 2) return 0;
  if (e->cmdStamp.size() != (size_t)e->G) { e->cmdStamp.assign((size_t)e->G, 0u); e->cmdSeen.assign((size_t)e->G, 0); e->cmdEpoch = 0; }
  if (++e->cmdEpoch == 0) { std::fill(e->cmdStamp.begin(), e->cmdStamp.end(), 0u); e->cmdEpoch = 1; }
  for (size_t i = 0; i < n; i++) {
    const int g = groups[i * stride];
    if (e->cmdStamp[g] == e->cmdEpoch) { level[i] = ++e->cmdSeen[g]; maxLevel = std::max(maxLevel, level[i]); }
    else { e->cmdStamp[g] = e->cmdEpoch; e->cmdSeen[g] = 0; }
  }
  return maxLevel;
}

static int ensure_cmd_buf(rg_engine *e, size_t n) {
  if (n <= e->dCmdCap) return RG_OK;
  if (e->dCmd) { hipFree(e->dCmd); e->dCmd = nullptr; }
  if (e->hCmd) { hipHostFree(e->hCmd); e->hCmd = nullptr; }
  e->dCmdCap = 0;
  size_t cap = std::max<size_t>(n, 1024);
  if (hipMalloc((void **)&e->dCmd, cap * sizeof(HostCmd)) != hipSuccess) return RG_ERR_NOMEM;
  if (hipHostMalloc((void **)&e->hCmd, cap * sizeof(HostCmd), hipHostMallocDefault) != hipSuccess) { hipFree(e->dCmd); e->dCmd = nullptr; return RG_ERR_NOMEM; }
  e->dCmdCap = cap;
  return RG_OK;
}

// Runs the n host commands the caller has written into e->hCmd (pinned; ensure_cmd_buf(e, n) first), in order, and
// leaves their results there.  Commands on one group are kept in order by cutting the batch into launches in which
// every group appears at most once; a batch that names every group once -- the common one -- is one launch, sent and
// fetched in place.
static int run_staged_cmds(rg_engine *e, size_t n) {
  if (n == 0) return RG_OK;
  HostCmd *cmds = e->hCmd;
  for (size_t i = 0; i < n; i++) if (cmds[i].group < 0 || cmds[i].group >= e->G || cmds[i].peer < 0 || cmds[i].peer >= e->P) return RG_ERR_RANGE;
  int rc = settle_client(e);           // host verbs neither read nor write the lazy wake-up times (make_peer overwrites them)
  if (rc == RG_OK) rc = run_derive(e);
  if (rc != RG_OK) return rc;
  std::vector<int> level;
  const int maxLevel = batch_levels(e, &cmds[0].group, sizeof(HostCmd) / sizeof(int32_t), n, level);
  if (maxLevel == 0) {
    HIPCHK(hipMemcpyAsync(e->dCmd, cmds, n * sizeof(HostCmd), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_host_cmds, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, e->stream, e->S, e->dc, e->PP, e->dCmd, (int)n, (int)e->now * RG_TICK_MS);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(cmds, e->dCmd, n * sizeof(HostCmd), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RG_OK;
  }
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
  int rc = ensure_cmd_buf(e, 1);
  if (rc != RG_OK) return rc;
  memset(&e->hCmd[0], 0, sizeof(HostCmd));
  e->hCmd[0].op = op; e->hCmd[0].group = g; e->hCmd[0].peer = p; e->hCmd[0].arg = arg;
  rc = run_staged_cmds(e, 1);
  if (rc == RG_OK && res) *res = e->hCmd[0];
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

static int ensure_start_buf(rg_engine *e, size_t n) {
  if (n <= e->startCap) return RG_OK;
  if (e->hIn) hipHostFree(e->hIn);
  if (e->hOut) hipHostFree(e->hOut);
  e->dIn = nullptr; e->dOut = nullptr; e->hIn = nullptr; e->hOut = nullptr; e->startCap = 0;
  const size_t cap = std::max<size_t>(n, 1024);
  // pinned host memory the device reads and writes in place (dIn / dOut: the device's addresses of the same buffers)
  if (hipHostMalloc((void **)&e->hIn, cap * sizeof(StartIn), hipHostMallocMapped) != hipSuccess ||
      hipHostMalloc((void **)&e->hOut, cap * sizeof(StartOut), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void **)&e->dIn, e->hIn, 0) != hipSuccess ||
      hipHostGetDevicePointer((void **)&e->dOut, e->hOut, 0) != hipSuccess) return RG_ERR_NOMEM;   // (freed by the next call or rg_destroy)
  e->startCap = cap;
  return RG_OK;
}

extern "C" int rg_start_batch(rg_engine *e, rg_start_req *reqs, int64_t n) {
  LOCK(e);
  if (n < 0 || (n > 0 && !reqs)) return RG_ERR_ARG;
  if (n == 0) return RG_OK;
  // The common batch -- every group named at most once -- goes in one pass over the requests: 16 bytes per Start to the
  // device, one kernel, 16 bytes back.  A batch that names a group twice takes the general path below, which keeps the
  // calls on one group in order.
  if (n <= 0x7fffffff && ensure_start_buf(e, (size_t)n) == RG_OK) {
    if (e->cmdStamp.size() != (size_t)e->G) { e->cmdStamp.assign((size_t)e->G, 0u); e->cmdSeen.assign((size_t)e->G, 0); e->cmdEpoch = 0; }
    if (++e->cmdEpoch == 0) { std::fill(e->cmdStamp.begin(), e->cmdStamp.end(), 0u); e->cmdEpoch = 1; }
    bool once = true;
    for (int64_t i = 0; i < n; i++) {
      const int g = reqs[i].group, p = reqs[i].peer;
      if (g < 0 || g >= e->G || p < 0 || p >= e->P) return RG_ERR_RANGE;
      if (e->cmdStamp[g] == e->cmdEpoch) { once = false; break; }
      e->cmdStamp[g] = e->cmdEpoch;
      e->hIn[i].group = g; e->hIn[i].peer = p; e->hIn[i].cmd = reqs[i].command;
    }
    if (once) {
      int rs = settle_client(e);
      if (rs == RG_OK) rs = run_derive(e);
      if (rs != RG_OK) return rs;
      // the kernel reads the requests from, and writes the answers to, the pinned host buffers themselves: one operation on
      // the stream instead of copy + kernel + copy
      hipLaunchKernelGGL(k_start_batch, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, e->stream, e->S, e->dc, e->PP, e->dIn, e->dOut, (int)n);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(e->stream));
      for (int64_t i = 0; i < n; i++) { reqs[i].index = e->hOut[i].index; reqs[i].term = e->hOut[i].term; reqs[i].is_leader = e->hOut[i].isLeader; }
      return RG_OK;
    }
  }
  int rc = ensure_cmd_buf(e, (size_t)n);
  if (rc != RG_OK) return rc;
  HostCmd *v = e->hCmd;
  for (int64_t i = 0; i < n; i++) { v[i].op = OP_START; v[i].group = reqs[i].group; v[i].peer = reqs[i].peer; v[i].pad = 0; v[i].arg = reqs[i].command; v[i].r0 = v[i].r1 = v[i].r2 = v[i].r3 = 0; }
  rc = run_staged_cmds(e, (size_t)n);
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
extern "C" int rg_kill(rg_engine *e, int32_t g, int32_t p) { LOCK(e); return run_one(e, OP_KILL, g, p, 0, nullptr); }
extern "C" int rg_restart(rg_engine *e, int32_t g, int32_t p) {
  LOCK(e);
  return run_one(e, OP_RESTART, g, p, 0, nullptr);   // the new instance replays from index 1 (raft.go:717) into the same apply ring
}

extern "C" int rg_set_unreliable(rg_engine *e, int32_t a, int32_t b, int32_t d) {
  LOCK(e);
  if (a < 0 || b < 0 || d < 0 || a > 1000 || b > 1000 || d > 1000) return RG_ERR_ARG;
  e->cfg.req_drop_permille = a; e->cfg.reply_drop_permille = b; e->cfg.reply_delay_permille = d;
  fill_net(e);
  return RG_OK;
}

struct TmpBuf {
  std::vector<void *> p;
  ~TmpBuf() { for (void *q : p) hipFree(q); }
  template <typename T> T *get(size_t n) { void *q = nullptr; if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr; p.push_back(q); return (T *)q; }
};

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

// Network.Reliable / LongReordering of ONE tester config (config.go:288-294: every config owns its labrpc.Network).
extern "C" int rg_set_unreliable_group(rg_engine *e, int32_t g, int32_t a, int32_t b, int32_t d) {
  LOCK(e);
  if (a < 0 || b < 0 || d < 0 || a > 1000 || b > 1000 || d > 1000) return RG_ERR_ARG;
  if (g < 0 || g >= e->G) return RG_ERR_RANGE;
  if (!e->netCfg) {
    int rc = dalloc(e, &e->netCfg, (size_t)e->Gpad);
    if (rc != RG_OK) return rc;
    HIPCHK(hipMemsetAsync(e->netCfg, 0, (size_t)e->Gpad * sizeof(unsigned), e->stream));
    e->H.netCfg = e->netCfg;
  }
  const unsigned w = 0x80000000u | (unsigned)a | ((unsigned)b << 10) | ((unsigned)d << 20);
  HIPCHK(hipMemcpyAsync(e->netCfg + g, &w, 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_set_schedule(rg_engine *e, const rg_sched_op *ops, int64_t n) {
  LOCK(e);
  if (n < 0 || n > (1 << 24) || (n > 0 && !ops)) return RG_ERR_ARG;
  for (int64_t i = 0; i < n; i++) {
    if (ops[i].op < RG_OP_CONNECT || ops[i].op >= RG_OP_END || ops[i].tick < 0 || (i > 0 && ops[i].tick < ops[i - 1].tick)) return RG_ERR_ARG;
    const int a = ops[i].a;
    const bool isReg = ops[i].op == RG_OP_START_ALL || (ops[i].op >= RG_OP_PICK_LEADER && ops[i].op <= RG_OP_PICK_OTHER);
    if (isReg ? (a < 0 || a > 3) : !((a >= 0 && a < e->P) || a == RG_SEL_ALL || ((a & ~0x3F) == 0x100 && (a & 15) < e->P)) && ops[i].op != RG_OP_START_ANY) return RG_ERR_ARG;
  }
  { int rs = settle(e); if (rs != RG_OK) return rs; }
  e->wakeDirty = true;                               // the schedule's ticks are events
  HIPCHK(hipStreamSynchronize(e->stream));
  e->H.sched = nullptr; e->H.schedN = 0; e->schedN = 0;
  e->schedHost.assign(ops, ops + n);
  int first = 0;                                   // ops before the current tick are never executed
  while (first < n && ops[first].tick < e->now) first++;
  if (n > 0) {
    if ((size_t)n > e->schedCap) {                 // (a smaller earlier buffer stays until rg_destroy)
      const size_t cap = std::max<size_t>((size_t)n, 2 * e->schedCap);
      int rc = dalloc(e, &e->sched, cap);
      if (rc != RG_OK) { e->schedCap = 0; return rc; }
      e->schedCap = cap;
    }
    HIPCHK(hipMemcpyAsync(e->sched, ops, (size_t)n * sizeof(rg_sched_op), hipMemcpyHostToDevice, e->stream));
    e->H.sched = e->sched; e->H.schedN = (int)n; e->schedN = (int)n;
  }
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->S.schedPos, first, (size_t)e->Gpad, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

// make_config (config.go:60-96) for groups of a running engine: fresh peers (empty Persisters, term 0, all
// connected, side 0), fresh tester record and flags, the engine's net rates, booted at the current tick.
extern "C" int rg_reset_groups(rg_engine *e, const int32_t *groups, int64_t n) {
  LOCK(e);
  if (n < 0 || (n > 0 && !groups)) return RG_ERR_ARG;
  if (n == 0) return RG_OK;
  for (int64_t i = 0; i < n; i++) if (groups[i] < 0 || groups[i] >= e->G) return RG_ERR_RANGE;
  { std::vector<int> level; if (batch_levels(e, groups, 1, (size_t)n, level) > 0) return RG_ERR_ARG; }   // one block per group: no repeats
  int rc = settle_client(e);             // client Starts the old peers still owe are carried out (and counted) first
  if (rc == RG_OK) rc = run_derive(e);
  if (rc != RG_OK) return rc;
  TmpBuf tb;
  int *dg = tb.get<int>((size_t)n);
  if (!dg) return RG_ERR_NOMEM;
  HIPCHK(hipMemcpyAsync(dg, groups, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_reset_groups, dim3((unsigned)n), dim3(256), 0, e->stream, e->S, e->dc, e->PP, dg, (int)e->now);
  HIPCHK(hipGetLastError());
  const unsigned zero = 0;
  if (e->netCfg) for (int64_t i = 0; i < n; i++) HIPCHK(hipMemcpyAsync(e->netCfg + groups[i], &zero, 4, hipMemcpyHostToDevice, e->stream));
  if (e->applyCap > 0) {               // a new applyCh per peer: what the old peers emitted and nobody read is gone
    std::vector<unsigned> heads;
    if (n > 8) {                       // many groups: the whole array in one copy
      heads.resize((size_t)e->nLanes);
      rc = d2h(e, heads.data(), e->applyHead, heads.size() * sizeof(unsigned));
      if (rc != RG_OK) return rc;
      for (int64_t i = 0; i < n; i++)
        for (int p = 0; p < e->P; p++) e->drained[(size_t)groups[i] * e->P + p] = heads[(size_t)groups[i] * e->PP + p];
    } else {
      heads.resize((size_t)e->PP);
      for (int64_t i = 0; i < n; i++) {
        rc = d2h(e, heads.data(), e->applyHead + (size_t)groups[i] * e->PP, (size_t)e->P * sizeof(unsigned));
        if (rc != RG_OK) return rc;
        for (int p = 0; p < e->P; p++) e->drained[(size_t)groups[i] * e->P + p] = heads[p];
      }
    }
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_drain_applied(rg_engine *e, int32_t g, int32_t p, rg_apply_msg *buf, int64_t cap, int64_t *n_out) {
  LOCK(e);
  if (!n_out || cap < 0 || (cap > 0 && !buf)) return RG_ERR_ARG;
  if (g < 0 || g >= e->G || p < 0 || p >= e->P) return RG_ERR_RANGE;
  *n_out = 0;
  if (e->applyCap <= 0) return RG_ERR_STATE;          // the engine was created without an apply ring
  unsigned head = 0;
  int rc = d2h(e, &head, e->applyHead + (size_t)g * e->PP + p, 4);
  if (rc != RG_OK) return rc;
  unsigned &cur = e->drained[(size_t)g * e->P + p];
  int status = RG_OK;
  if (head - cur > (unsigned)e->applyCap) {           // the peer emitted more than the ring holds since the last drain
    cur = head - (unsigned)e->applyCap;               // the oldest message still there
    status = RG_ERR_OVERRUN;
  }
  int64_t n = (int64_t)(head - cur);
  if (n > cap) n = cap;
  const rg_apply_msg *ring = e->applyRing + ((size_t)g * e->P + p) * (size_t)e->applyCap;
  int64_t done = 0;
  while (done < n) {                                  // at most two stretches: the ring wraps
    const unsigned slot = (cur + (unsigned)done) % (unsigned)e->applyCap;
    int64_t run = (int64_t)e->applyCap - slot;
    if (run > n - done) run = n - done;
    rc = d2h(e, buf + done, ring + slot, (size_t)run * sizeof(rg_apply_msg));
    if (rc != RG_OK) return rc;
    done += run;
  }
  cur += (unsigned)n;
  *n_out = n;
  return status;
}

extern "C" int rg_snapshot_export(rg_engine *e, int32_t g, int32_t p, void *buf, int64_t cap, int64_t *size_out) {
  LOCK(e);
  if (g < 0 || g >= e->G || p < 0 || p >= e->P) return RG_ERR_RANGE;
  { int rs = settle_client(e); if (rs != RG_OK) return rs; }
  const size_t gl = (size_t)g * e->PP + p;
  int flags = 0, hdr[3];
  HIPCHK(hipMemcpyAsync(&flags, e->S.flags + gl, 4, hipMemcpyDeviceToHost, e->stream));     // four words, one wait
  HIPCHK(hipMemcpyAsync(&hdr[0], e->S.snapTerm + gl, 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(&hdr[1], e->S.snapVote + gl, 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(&hdr[2], e->S.snapLen + gl, 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  int rc = RG_OK;
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
    std::vector<char> rows(host_span(e, n));
    const char *srcRows = (((unsigned)flags & F_SNAPEXT) && e->S.snapRows) ? e->S.snapRows : e->S.logRows;
    rc = d2h(e, rows.data(), srcRows + (size_t)g * e->groupBytes, rows.size());
    if (rc != RG_OK) return rc;
    for (int i = 0; i < n; i++) {
      char *o = (char *)buf + 16 + 16 * (size_t)i;
      memcpy(o, rows.data() + host_toff(e, i, p), 4);
      memset(o + 4, 0, 4);
      memcpy(o + 8, rows.data() + host_coff(e, i, p), 8);
    }
  }
  return RG_OK;
}

extern "C" int rg_snapshot_import(rg_engine *e, int32_t g, int32_t p, const void *buf, int64_t size) {
  LOCK(e);
  if (g < 0 || g >= e->G || p < 0 || p >= e->P) return RG_ERR_RANGE;
  const size_t gl = (size_t)g * e->PP + p;
  int flags = 0;
  int rc = settle(e);
  if (rc != RG_OK) return rc;
  e->wakeDirty = true;
  rc = d2h(e, &flags, e->S.flags + gl, 4);
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
    std::vector<char> rows(host_span(e, n));
    char *dbase = e->S.snapRows + (size_t)g * e->groupBytes;
    rc = d2h(e, rows.data(), dbase, rows.size());
    if (rc != RG_OK) return rc;
    for (int i = 0; i < n; i++) {
      const char *s = (const char *)buf + 16 + 16 * (size_t)i;
      memcpy(rows.data() + host_toff(e, i, p), s, 4);
      memcpy(rows.data() + host_coff(e, i, p), s + 8, 8);
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
static void level_order(rg_engine *e, const int32_t *groups, int64_t n, std::vector<int> &order, std::vector<size_t> &startOf) {
  std::vector<int> level;
  const int maxLevel = batch_levels(e, groups, 1, (size_t)n, level);
  startOf.assign(maxLevel + 2, 0);
  for (int64_t i = 0; i < n; i++) startOf[level[i] + 1]++;
  for (int lv = 0; lv <= maxLevel; lv++) startOf[lv + 1] += startOf[lv];
  order.assign((size_t)n, 0);
  std::vector<size_t> at(startOf.begin(), startOf.end() - 1);
  for (int64_t i = 0; i < n; i++) order[at[level[i]]++] = (int)i;       // stable: requests of a level keep their order
}

extern "C" int rg_rpc_request_vote(rg_engine *e, int64_t n, const int32_t *groups, const int32_t *peers,
                                   const rg_vote_args *args, rg_vote_reply *replies) {
  LOCK(e);
  if (n < 0 || (n > 0 && (!groups || !peers || !args || !replies))) return RG_ERR_ARG;
  if (n == 0) return RG_OK;
  for (int64_t i = 0; i < n; i++) if (groups[i] < 0 || groups[i] >= e->G || peers[i] < 0 || peers[i] >= e->P) return RG_ERR_RANGE;
  int rc = settle(e);
  if (rc == RG_OK) rc = run_derive(e);
  if (rc != RG_OK) return rc;
  std::vector<int> order; std::vector<size_t> startOf;
  level_order(e, groups, n, order, startOf);
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
  int rc = settle(e);
  if (rc == RG_OK) rc = run_derive(e);
  if (rc != RG_OK) return rc;
  std::vector<int> order; std::vector<size_t> startOf;
  level_order(e, groups, n, order, startOf);
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

static int run_derive_if_dirty(rg_engine *e);
// ---- bulk state access -----------------------------------------------------------
static int64_t field_bytes(const rg_engine *e, int field) {
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
  case RG_F_SCHEDPOS: return S.schedPos; case RG_F_SCHEDREGS: return S.schedRegs;
  default: return nullptr;
  }
}

static int rw_state(rg_engine *e, int field, void *buf, int64_t bytes, bool wr) {
  if (!buf) return RG_ERR_ARG;
  { int rs = settle(e); if (rs != RG_OK) return rs; }
  if (wr) e->wakeDirty = true;
  if (field >= RG_F_CACHE_LASTTERM && field <= RG_F_CACHE_COMMITTERM) {
    if (wr || bytes != (int64_t)e->G * e->P * 4) return RG_ERR_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    int rcd = run_derive_if_dirty(e);
    if (rcd != RG_OK) return rcd;
    const int *dev = field == RG_F_CACHE_LASTTERM ? e->S.lastTerm : field == RG_F_CACHE_FIRSTOWN ? e->S.firstOwn : e->S.commitTerm;
    std::vector<int> tmp((size_t)e->nLanes);
    int rcc = d2h(e, tmp.data(), dev, tmp.size() * 4);
    if (rcc != RG_OK) return rcc;
    for (int g = 0; g < e->G; g++) for (int p = 0; p < e->P; p++) ((int32_t *)buf)[(size_t)g * e->P + p] = tmp[(size_t)g * e->PP + p];
    return RG_OK;
  }
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
      if (field == RG_F_DEAD) {                      // 0 alive, 1 crashed, 2 killed: two bits of the flag word
        if (!wr) hv = ((unsigned)wv & F_DEAD) ? (((unsigned)wv & F_ZOMBIE) ? 2 : 1) : 0;
        else wv = (int)(((unsigned)wv & ~(F_DEAD | F_ZOMBIE)) | (hv ? F_DEAD : 0u) | (hv == 2 ? F_ZOMBIE : 0u));
        continue;
      }
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
    const size_t gB = e->groupBytes;
    const int chunk = std::max(1, (int)std::min<int64_t>(G, (64ll << 20) / (int64_t)gB));
    std::vector<char> rows((size_t)chunk * gB), rows2(both ? rows.size() : 0);
    for (int g0 = 0; g0 < G; g0 += chunk) {
      const int ng = std::min(chunk, G - g0);
      const size_t nb = (size_t)ng * gB;
      rc = d2h(e, rows.data(), dev + (size_t)g0 * gB, nb);
      if (rc != RG_OK) return rc;
      if (both) { rc = d2h(e, rows2.data(), e->S.snapRows + (size_t)g0 * gB, nb); if (rc != RG_OK) return rc; }
      for (int g = 0; g < ng; g++) for (int p = 0; p < P; p++) {
        const bool ext = both && (((unsigned)flags[(size_t)(g0 + g) * PP + p]) & F_SNAPEXT);
        char *base = ext ? rows2.data() : rows.data();
        for (int i = 0; i < L; i++) {
          char *grp = base + (size_t)g * gB;
          const size_t hi = ((size_t)(g0 + g) * P + p) * L + i;
          if (isCmd) { u64 *hv = (u64 *)buf + hi; char *w = grp + host_coff(e, i, p); if (wr) memcpy(w, hv, 8); else memcpy(hv, w, 8); }
          else { int32_t *hv = (int32_t *)buf + hi; char *w = grp + host_toff(e, i, p); if (wr) memcpy(w, hv, 4); else memcpy(hv, w, 4); }
        }
      }
      if (wr) { rc = h2d(e, dev + (size_t)g0 * gB, rows.data(), nb); if (rc != RG_OK) return rc; }
    }
    if (wr) { e->derivedDirty = true; e->syncDirty = true; }
    return RG_OK;
  }
  if (field >= RG_F_STARTSEQ && field <= RG_F_SCHEDREGS) {
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

static int run_derive_if_dirty(rg_engine *e) { return run_derive(e); }
extern "C" int rg_read_state(rg_engine *e, int32_t field, void *dst, int64_t bytes) { LOCK(e); return rw_state(e, field, dst, bytes, false); }
extern "C" int rg_write_state(rg_engine *e, int32_t field, const void *src, int64_t bytes) { LOCK(e); return rw_state(e, field, (void *)src, bytes, true); }

extern "C" int rg_state_hash(rg_engine *e, uint64_t *out, int64_t n) {
  LOCK(e);
  if (!out || n != e->G) return RG_ERR_ARG;
  { int rs = settle(e); if (rs != RG_OK) return rs; }
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
  int rc = settle_client(e);
  if (rc == RG_OK) rc = d2h(e, out, e->S.counters, RG_N_COUNTERS * 8);
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
  { int rs = settle_client(e); if (rs != RG_OK) return rs; }
  const long long n = (long long)e->G * e->L;
  hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->S, e->dc, e->PP);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return RG_OK;
}

extern "C" int rg_safety_flags(rg_engine *e, uint32_t *out) {
  LOCK(e);
  if (!out) return RG_ERR_ARG;
  { int rs = settle_client(e); if (rs != RG_OK) return rs; }
  HIPCHK(hipMemsetAsync(e->S.flagSummary, 0, 8, e->stream));
  hipLaunchKernelGGL(k_flag_summary, dim3((e->G + 255) / 256), dim3(256), 0, e->stream, e->S, e->G);
  HIPCHK(hipGetLastError());
  return d2h(e, out, e->S.flagSummary, 8);
}

// Packs counters and safety summary of this engine's groups into redBuf (on the engine's stream).
static int pack_summary(rg_engine *e) {
  { int rs = settle_client(e); if (rs != RG_OK) return rs; }
  HIPCHK(hipMemsetAsync(e->redBuf, 0, (RG_N_COUNTERS + RG_N_SUMMARY) * sizeof(u64), e->stream));
  hipLaunchKernelGGL(k_summary, dim3((e->G + 255) / 256), dim3(256), 0, e->stream, e->S, e->G, e->P, e->PP,
                     (unsigned long long)e->G * (unsigned long long)e->now, (unsigned long long *)e->redBuf);
  HIPCHK(hipGetLastError());
  return RG_OK;
}

extern "C" int rg_safety_summary(rg_engine *e, uint64_t *out) {
  LOCK(e);
  if (!out) return RG_ERR_ARG;
  int rc = pack_summary(e);
  if (rc != RG_OK) return rc;
  return d2h(e, out, e->redBuf + RG_N_COUNTERS, RG_N_SUMMARY * sizeof(u64));
}

// ---------------------------------------------------------------- RCCL, loaded when first needed
// The declarations below restate the four entry points used, as rccl.h publishes them; nothing is linked.
typedef struct { char internal[RG_COMM_ID_BYTES]; } rg_nccl_id;      // ncclUniqueId: 128 opaque bytes, passed by value
struct Rccl {
  void *lib = nullptr;
  int (*getUniqueId)(rg_nccl_id *) = nullptr;
  int (*commInitRank)(void **comm, int nranks, rg_nccl_id id, int rank) = nullptr;
  int (*commDestroy)(void *comm) = nullptr;
  int (*allReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream) = nullptr;
  const char *(*errorString)(int) = nullptr;
};
enum { RG_NCCL_SUM = 0, RG_NCCL_UINT64 = 5 };                        // ncclSum, ncclUint64
static const Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define RCCL_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.lib, #sym)
    RCCL_SYM(getUniqueId, ncclGetUniqueId); RCCL_SYM(commInitRank, ncclCommInitRank); RCCL_SYM(commDestroy, ncclCommDestroy);
    RCCL_SYM(allReduce, ncclAllReduce); RCCL_SYM(errorString, ncclGetErrorString);
#undef RCCL_SYM
    if (!(r.getUniqueId && r.commInitRank && r.commDestroy && r.allReduce)) r.lib = nullptr;
  });
  return r.lib ? &r : nullptr;
}
static int rccl_fail(const Rccl *r, const char *what, int st) {
  fprintf(stderr, "raftgpu: %s: %s\n", what, r->errorString ? r->errorString(st) : "RCCL error");
  return RG_ERR_HIP;
}

static void comm_release(void *comm) { const Rccl *r = rccl(); if (r) r->commDestroy(comm); }

extern "C" int rg_comm_unique_id(void *id_out) {
  if (!id_out) return RG_ERR_ARG;
  const Rccl *r = rccl();
  if (!r) return RG_ERR_STATE;
  rg_nccl_id id;
  const int st = r->getUniqueId(&id);
  if (st != 0) return rccl_fail(r, "ncclGetUniqueId", st);
  memcpy(id_out, &id, sizeof id);
  return RG_OK;
}

extern "C" int rg_comm_init(rg_engine *e, int32_t rank, int32_t n_ranks, const void *idp) {
  LOCK(e);
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !idp)) return RG_ERR_ARG;
  if (e->comm) return RG_ERR_STATE;
  const Rccl *r = rccl();
  if (!r) return RG_ERR_STATE;
  rg_nccl_id id;
  if (idp) memcpy(&id, idp, sizeof id);
  else { const int st = r->getUniqueId(&id); if (st != 0) return rccl_fail(r, "ncclGetUniqueId", st); }
  void *comm = nullptr;
  const int st = r->commInitRank(&comm, n_ranks, id, rank);
  if (st != 0 || !comm) return rccl_fail(r, "ncclCommInitRank", st);
  e->comm = comm;
  return RG_OK;
}

extern "C" int rg_comm_destroy(rg_engine *e) {
  LOCK(e);
  if (!e->comm) return RG_OK;
  const Rccl *r = rccl();
  if (!r) return RG_ERR_STATE;
  HIPCHK(hipStreamSynchronize(e->stream));
  const int st = r->commDestroy(e->comm);
  e->comm = nullptr;
  return st == 0 ? RG_OK : rccl_fail(r, "ncclCommDestroy", st);
}

extern "C" int rg_allreduce_summary(rg_engine *e, void *nccl_comm, uint64_t *counters_out, uint64_t *summary_out) {
  LOCK(e);
  int rc = pack_summary(e);
  if (rc != RG_OK) return rc;
  void *comm = nccl_comm ? nccl_comm : e->comm;
  if (comm) {
    const Rccl *r = rccl();
    if (!r) return RG_ERR_STATE;
    const int st = r->allReduce(e->redBuf, e->redBuf, (size_t)(RG_N_COUNTERS + RG_N_SUMMARY), RG_NCCL_UINT64, RG_NCCL_SUM, comm, e->stream);
    if (st != 0) return rccl_fail(r, "ncclAllReduce", st);
  }
  u64 host[RG_N_COUNTERS + RG_N_SUMMARY];
  rc = d2h(e, host, e->redBuf, sizeof host);
  if (rc != RG_OK) return rc;
  if (counters_out) memcpy(counters_out, host, RG_N_COUNTERS * sizeof(u64));
  if (summary_out) memcpy(summary_out, host + RG_N_COUNTERS, RG_N_SUMMARY * sizeof(u64));
  return RG_OK;
}

extern "C" int rg_device_buffers(rg_engine *e, void **counters_dev, void **flag_summary_dev) {
  LOCK(e);
  if (counters_dev) *counters_dev = e->S.counters;
  if (flag_summary_dev) *flag_summary_dev = e->S.flagSummary;
  return RG_OK;
}

#if defined(RG_SECTION_COUNTS) || defined(RG_STAMPS) || defined(RG_DSTAMPS)
// profiling builds only: how often each section of k_tick ran, per wave
extern "C" int rg_debug_section_counts(rg_engine *e, uint64_t *out /* RG_CTR_SLOTS */) {
  LOCK(e);
  if (!out) return RG_ERR_ARG;
  return d2h(e, out, e->S.counters, RG_CTR_SLOTS * 8);
}
#endif
