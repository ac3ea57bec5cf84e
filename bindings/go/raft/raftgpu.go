// Package raft: the reference's `package raft` surface (src/raft/raft.go:42-46,90-100,161-195,344-364,
// 377-385,700-731) served by the batched engine of include/raftgpu.h through cgo.
//
// This file replaces src/raft/raft.go; src/raft/persister.go, config.go and test_test.go stay as they are
// except for the one-line calls into config_hooks.go.  labrpc_shim.go gives the exported handlers
// (RequestVote, AppendEntries) the shape of a labrpc end.  There is no Go toolchain in the image this
// repository is built in, so the package is not compiled there; the same calls, in the same order, are
// exercised through ctypes by raft_amd/raft.py, raft_amd/harness.py and the tests.
package raft

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../raft_amd/lib -lraftgpu
#include "raftgpu.h"
*/
import "C"

import (
	"sync"
	"sync/atomic"
	"time"
	"unsafe"

	"../labrpc"
)

// ApplyMsg, raft.go:42-46.
type ApplyMsg struct {
	CommandValid bool
	Command      interface{}
	CommandIndex int
}

// RPC argument and reply types, raft.go:161-195 (field for field; used by labrpc_shim.go).
type Entry struct {
	Term    int32
	Command interface{}
}
type RequestVoteArgs struct {
	Term         int32
	CandidateId  int
	LastLogIndex int
	LastLogTerm  int32
}
type RequestVoteReply struct {
	Term        int32
	VoteGranted bool
}
type AppendEntriesArgs struct {
	Term         int32
	LeaderId     int
	PrevLogIndex int
	PrevLogTerm  int32
	Entries      []Entry
	LeaderCommit int
}
type AppendEntriesReply struct {
	Term               int32
	Success            bool
	ConflictTerm       int32
	ConflictFirstIndex int
}

const (
	engineGroups = 4096 // tester configs one engine can hold at a time
	engineLogCap = 4096
)

// engine is the process-wide engine for one peer count: every tester config (config.go:60-96) is one of its
// groups; ONE goroutine steps it for all of them.
type engine struct {
	mu     sync.Mutex
	e      *C.rg_engine
	peers  int
	free   []C.int32_t       // group slots not in use
	groups map[C.int32_t]*Group
}

var (
	enginesMu sync.Mutex
	engines   = map[int]*engine{} // by number of peers
	current   *Group              // the config whose peers are being made (make_config is not re-entrant)
)

func status(rc C.int, what string) {
	if rc != C.RG_OK {
		panic(what + ": " + C.GoString(C.rg_strerror(rc)))
	}
}

func engineFor(n int) *engine {
	enginesMu.Lock()
	defer enginesMu.Unlock()
	if en, ok := engines[n]; ok {
		return en
	}
	var c C.rg_config
	c.n_groups, c.n_peers, c.log_cap = C.int32_t(engineGroups), C.int32_t(n), C.int32_t(engineLogCap)
	c.apply_ring_cap = C.int32_t(engineLogCap) // applyCh is read from here (rg_drain_applied)
	c.account_bytes = 1
	c.seed = C.uint64_t(time.Now().UnixNano()) // config.go:30-35 seeds from crypto/rand
	en := &engine{peers: n, groups: map[C.int32_t]*Group{}}
	status(C.rg_create(&c, &en.e), "rg_create")
	// rg_create boots every group; a slot nobody has taken must not elect, heartbeat or count RPCs (RPCTotal and
	// BytesTotal read engine-wide counters): all peers crashed and off the net until BeginConfig's rg_reset_groups
	// makes a slot's peers afresh.
	parked := make([]C.int32_t, engineGroups*n)
	status(C.rg_write_state(en.e, C.RG_F_CONNECTED, unsafe.Pointer(&parked[0]), C.int64_t(4*len(parked))), "rg_write_state(connected)")
	for i := range parked {
		parked[i] = 1
	}
	status(C.rg_write_state(en.e, C.RG_F_DEAD, unsafe.Pointer(&parked[0]), C.int64_t(4*len(parked))), "rg_write_state(dead)")
	for g := engineGroups - 1; g >= 0; g-- {
		en.free = append(en.free, C.int32_t(g))
	}
	engines[n] = en
	go en.drive()
	return en
}

// Group is one Raft group of the engine = one tester config.
type Group struct {
	mu        sync.Mutex
	en        *engine
	g         C.int32_t
	n         int
	cmds      map[uint64]interface{} // commands are opaque to the engine: it carries 64-bit handles
	records   map[string][]byte      // gob raftstate this config wrote -> the engine record it was made from
	next      uint64
	rafts     []*Raft
	connected []bool
	rpcs0     uint64 // engine-wide counters when the config began (one config runs at a time in `go test`)
	bytes0    uint64
}

// BeginConfig is make_config's part (config.go:60-96): a group slot with fresh peers -- Make with empty
// Persisters at the engine's current tick -- and the config's own network rates.
func BeginConfig(n int, unreliable bool) *Group {
	en := engineFor(n)
	en.mu.Lock()
	defer en.mu.Unlock()
	if len(en.free) == 0 {
		panic("raftgpu: all group slots of the engine are in use")
	}
	g := en.free[len(en.free)-1]
	en.free = en.free[:len(en.free)-1]
	status(C.rg_reset_groups(en.e, &g, 1), "rg_reset_groups")
	grp := &Group{en: en, g: g, n: n, cmds: map[uint64]interface{}{}, records: map[string][]byte{}, next: 1,
		rafts: make([]*Raft, n), connected: make([]bool, n)}
	grp.setRates(unreliable, false)
	grp.rpcs0, grp.bytes0 = grp.counters()
	en.groups[g] = grp
	current = grp
	return grp
}

// EndConfig is cfg.cleanup's part (config.go:224-233): the slot goes back to the engine.
func (grp *Group) EndConfig() {
	en := grp.en
	en.mu.Lock()
	defer en.mu.Unlock()
	for i := 0; i < grp.n; i++ { // nobody steps, votes or leads in a slot that is not in use
		C.rg_crash(en.e, grp.g, C.int32_t(i))
	}
	delete(en.groups, grp.g)
	en.free = append(en.free, grp.g)
}

// SetConnected mirrors net.Enable for all ends of server i (config.go:235-278).
func (grp *Group) SetConnected(i int, on bool) {
	v := C.int32_t(0)
	if on {
		v = 1
	}
	grp.mu.Lock()
	grp.connected[i] = on
	grp.mu.Unlock()
	status(C.rg_set_connected(grp.en.e, grp.g, C.int32_t(i), v), "rg_set_connected")
}

// setRates mirrors net.Reliable / net.LongReordering of THIS config's network (config.go:288-298,
// labrpc.go:169-181, 226, 273, 276-285).
func (grp *Group) setRates(unreliable, longReordering bool) {
	drop, delay := C.int32_t(0), C.int32_t(0)
	if unreliable {
		drop = 100
	}
	if longReordering {
		delay = 667
	}
	status(C.rg_set_unreliable_group(grp.en.e, grp.g, drop, drop, delay), "rg_set_unreliable_group")
}

func (grp *Group) counters() (rpcs, bytes uint64) {
	var c [C.RG_N_COUNTERS]C.uint64_t
	status(C.rg_counters(grp.en.e, &c[0]), "rg_counters")
	var b C.uint64_t
	status(C.rg_rpc_bytes(grp.en.e, 8, &b), "rg_rpc_bytes")
	return uint64(c[C.RG_C_RPC_SENT]), uint64(b)
}

// RPCTotal / BytesTotal mirror net.GetTotalCount / GetTotalBytes (config.go:284-294) since BeginConfig.
func (grp *Group) RPCTotal() int {
	r, _ := grp.counters()
	return int(r - grp.rpcs0)
}
func (grp *Group) BytesTotal() int64 {
	_, b := grp.counters()
	return int64(b - grp.bytes0)
}

// Raft, raft.go:57-86: all state lives in the engine; the object is a handle.
type Raft struct {
	grp     *Group
	me      int
	dead    int32
	applyCh chan ApplyMsg
	ps      *Persister
}

// GetState, raft.go:90-100.
func (rf *Raft) GetState() (int, bool) {
	var term, lead C.int32_t
	status(C.rg_get_state(rf.grp.en.e, rf.grp.g, C.int32_t(rf.me), &term, &lead), "rg_get_state")
	return int(term), lead != 0 && !rf.killed()
}

// Start, raft.go:344-364.
func (rf *Raft) Start(command interface{}) (int, int, bool) {
	grp := rf.grp
	if rf.killed() {
		t, _ := rf.GetState()
		return -1, t, false
	}
	grp.mu.Lock()
	h := grp.next
	grp.next++
	grp.cmds[h] = command
	grp.mu.Unlock()
	var idx, term, lead C.int32_t
	status(C.rg_start(grp.en.e, grp.g, C.int32_t(rf.me), C.uint64_t(h), &idx, &term, &lead), "rg_start")
	return int(idx), int(term), lead != 0
}

// Kill, raft.go:377-385: the dead flag and nothing else -- the peer stays on the net (a Call to it still
// returns true, with the reply untouched).  The tester's crash1 takes it off the net first (hookCrash).
func (rf *Raft) Kill() {
	if atomic.CompareAndSwapInt32(&rf.dead, 0, 1) {
		status(C.rg_kill(rf.grp.en.e, rf.grp.g, C.int32_t(rf.me)), "rg_kill")
	}
}

// AllReduceSummary is the job's one cross-GPU exchange: this engine's counters (RG_C_*) and safety summary (RG_S_*),
// summed over every rank of the communicator made with CommInit -- one ncclAllReduce over xGMI inside the library.
// Every rank calls it at the same reporting boundaries.
func AllReduceSummary(n int) (counters [C.RG_N_COUNTERS]uint64, summary [C.RG_N_SUMMARY]uint64) {
	en := engineFor(n)
	var c [C.RG_N_COUNTERS]C.uint64_t
	var s [C.RG_N_SUMMARY]C.uint64_t
	status(C.rg_allreduce_summary(en.e, nil, &c[0], &s[0]), "rg_allreduce_summary")
	for i := range c {
		counters[i] = uint64(c[i])
	}
	for i := range s {
		summary[i] = uint64(s[i])
	}
	return
}

// CommUniqueId / CommInit: rank 0 makes the id, the host program carries its RG_COMM_ID_BYTES to the other ranks
// (one process per GPU), every rank then joins.
func CommUniqueId() (id [C.RG_COMM_ID_BYTES]byte) {
	status(C.rg_comm_unique_id(unsafe.Pointer(&id[0])), "rg_comm_unique_id")
	return
}
func CommInit(n, rank, nRanks int, id [C.RG_COMM_ID_BYTES]byte) {
	status(C.rg_comm_init(engineFor(n).e, C.int32_t(rank), C.int32_t(nRanks), unsafe.Pointer(&id[0])), "rg_comm_init")
}

func (rf *Raft) killed() bool { return atomic.LoadInt32(&rf.dead) == 1 }

// SyncPersister copies what persist() (raft.go:107-123) last stored for this peer -- it runs inside the
// engine -- into the peer's Persister.  The harness reads a Persister only when it crashes or restarts a
// server (config.go:110-126, 159-163), so that is when config_hooks.go calls this; nothing is exported per tick.
func (rf *Raft) SyncPersister() {
	if rf.killed() {
		return
	}
	var size C.int64_t
	e, g, me := rf.grp.en.e, rf.grp.g, C.int32_t(rf.me)
	status(C.rg_snapshot_export(e, g, me, nil, 0, &size), "rg_snapshot_export")
	if size == 0 {
		return
	}
	buf := make([]byte, int(size))
	status(C.rg_snapshot_export(e, g, me, unsafe.Pointer(&buf[0]), size, &size), "rg_snapshot_export")
	blob := rf.grp.gobFromRecord(buf) // the reference's own encoding (persist_gob.go)
	rf.grp.mu.Lock()
	rf.grp.records[string(blob)] = buf // a blob this config wrote comes back with its own command handles
	rf.grp.mu.Unlock()
	rf.ps.SaveRaftState(blob)
}

// Make, raft.go:700-731.  peers is kept for len(peers) only: the peers of a group talk inside the GPU.
// The first Make of a server in a config finds the fresh peer BeginConfig made; every later one (start1 after
// crash1) is Make + readPersist from the Persister's blob (raft.go:708-725).
func Make(peers []*labrpc.ClientEnd, me int, persister *Persister, applyCh chan ApplyMsg) *Raft {
	grp := current
	rf := &Raft{grp: grp, me: me, applyCh: applyCh, ps: persister}
	grp.mu.Lock()
	first := grp.rafts[me] == nil
	grp.mu.Unlock()
	blob := persister.ReadRaftState()
	if !first || len(blob) > 0 {
		grp.mu.Lock()
		rec, mine := grp.records[string(blob)]
		grp.mu.Unlock()
		if !mine && len(blob) > 0 { // written by other code, e.g. the reference's persist(): raft.go:116-122
			var err error
			if rec, err = grp.recordFromGob(blob); err != nil {
				rec = nil // readPersist logs and starts empty (raft.go:141-142)
			}
		}
		var p unsafe.Pointer
		if len(rec) > 0 {
			p = unsafe.Pointer(&rec[0])
		}
		status(C.rg_snapshot_import(grp.en.e, grp.g, C.int32_t(me), p, C.int64_t(len(rec))), "rg_snapshot_import")
		status(C.rg_restart(grp.en.e, grp.g, C.int32_t(me)), "rg_restart")
	}
	grp.mu.Lock()
	grp.rafts[me] = rf
	grp.mu.Unlock()
	return rf
}

// drive is the one long-running goroutine of an engine: ticks of 10 ms for every config it holds, then
// each live peer's ApplyMsgs, in the order the engine emitted them (raft.go:672-688).
func (en *engine) drive() {
	buf := make([]C.rg_apply_msg, 1024)
	for {
		time.Sleep(C.RG_TICK_MS * time.Millisecond)
		status(C.rg_step(en.e, 1), "rg_step")
		en.mu.Lock()
		groups := make([]*Group, 0, len(en.groups))
		for _, grp := range en.groups {
			groups = append(groups, grp)
		}
		en.mu.Unlock()
		for _, grp := range groups {
			grp.mu.Lock()
			rafts := append([]*Raft(nil), grp.rafts...)
			grp.mu.Unlock()
			for i, rf := range rafts {
				if rf == nil || rf.killed() {
					continue
				}
				for {
					var n C.int64_t
					rc := C.rg_drain_applied(en.e, grp.g, C.int32_t(i), &buf[0], C.int64_t(len(buf)), &n)
					if rc != C.RG_ERR_OVERRUN { // an overrun has been skipped over; what is left is delivered
						status(rc, "rg_drain_applied")
					}
					for k := 0; k < int(n); k++ {
						grp.mu.Lock()
						cmd := grp.cmds[uint64(buf[k].command)]
						grp.mu.Unlock()
						rf.applyCh <- ApplyMsg{true, cmd, int(buf[k].command_index)}
					}
					if int(n) < len(buf) {
						break
					}
				}
			}
		}
	}
}
