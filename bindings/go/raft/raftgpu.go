// Package raft: the reference's `package raft` surface (src/raft/raft.go:42-46,90-100,344-364,377-385,
// 700-731) served by the batched engine of include/raftgpu.h through cgo.
//
// This file replaces src/raft/raft.go; src/raft/persister.go, config.go and test_test.go stay as they
// are except for the hooks listed in hooks.go.  There is no Go toolchain in the image this
// repository is built in, so the file is not compiled there; the same calls, in the same order, are
// exercised through ctypes by raft_amd/raft.py and the tests.
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

// Group is one Raft group of the engine = one tester config (config.go:60-96).
type Group struct {
	mu     sync.Mutex
	e      *C.rg_engine
	g      C.int32_t
	n      int
	cmds   map[uint64]interface{} // commands are opaque to the engine: it carries 64-bit handles
	next   uint64
	rafts  []*Raft
	booted []bool
	stop   int32
}

// Raft, raft.go:57-86: all state lives in the engine; the object is a handle.
type Raft struct {
	grp     *Group
	me      int
	dead    int32
	applyCh chan ApplyMsg
	ps      *Persister
}

var (
	engMu   sync.Mutex
	eng     *C.rg_engine
	engCap  = 4096 // groups of the process-wide engine; a config takes the next free one
	engNext int
	current *Group // the config being built or run (tests of one package run one after the other)
)

func status(rc C.int, what string) {
	if rc != C.RG_OK {
		panic(what + ": " + C.GoString(C.rg_strerror(rc)))
	}
}

// BeginConfig is called from make_config (config.go:68) with the number of peers.
func BeginConfig(n int, logCap int) *Group {
	engMu.Lock()
	defer engMu.Unlock()
	if eng == nil || int(engPeers) != n {
		if eng != nil {
			C.rg_destroy(eng)
		}
		var c C.rg_config
		c.n_groups, c.n_peers, c.log_cap = C.int32_t(engCap), C.int32_t(n), C.int32_t(logCap)
		c.seed = C.uint64_t(time.Now().UnixNano())
		status(C.rg_create(&c, &eng), "rg_create")
		engPeers, engNext = C.int32_t(n), 0
	}
	grp := &Group{e: eng, g: C.int32_t(engNext), n: n, cmds: map[uint64]interface{}{}, next: 1,
		rafts: make([]*Raft, n), booted: make([]bool, n)}
	engNext++
	current = grp
	go grp.drive()
	return grp
}

var engPeers C.int32_t

// EndConfig is called from cfg.cleanup (config.go:224-233).
func (grp *Group) EndConfig() { atomic.StoreInt32(&grp.stop, 1) }

// SetConnected mirrors net.Enable for all ends of server i (config.go:235-278).
func (grp *Group) SetConnected(i int, on bool) {
	v := C.int32_t(0)
	if on {
		v = 1
	}
	status(C.rg_set_connected(grp.e, grp.g, C.int32_t(i), v), "rg_set_connected")
}

// SetUnreliable mirrors net.Reliable / net.LongReordering (config.go:288-298, labrpc.go:226,273,283-301).
func (grp *Group) SetUnreliable(unreliable, longReordering bool) {
	drop, delay := C.int32_t(0), C.int32_t(0)
	if unreliable {
		drop = 100
	}
	if longReordering {
		delay = 667
	}
	status(C.rg_set_unreliable(grp.e, drop, drop, delay), "rg_set_unreliable")
}

// RPCTotal mirrors net.GetTotalCount (config.go:284-286) for the whole engine.
func (grp *Group) RPCTotal() int {
	var c [C.RG_N_COUNTERS]C.uint64_t
	status(C.rg_counters(grp.e, &c[0]), "rg_counters")
	return int(c[C.RG_C_RPC_SENT])
}

// GetState, raft.go:90-100.
func (rf *Raft) GetState() (int, bool) {
	var term, lead C.int32_t
	status(C.rg_get_state(rf.grp.e, rf.grp.g, C.int32_t(rf.me), &term, &lead), "rg_get_state")
	return int(term), lead != 0 && !rf.killed()
}

// Start, raft.go:344-364.
func (rf *Raft) Start(command interface{}) (int, int, bool) {
	grp := rf.grp
	grp.mu.Lock()
	h := grp.next
	grp.next++
	grp.cmds[h] = command
	grp.mu.Unlock()
	var idx, term, lead C.int32_t
	status(C.rg_start(grp.e, grp.g, C.int32_t(rf.me), C.uint64_t(h), &idx, &term, &lead), "rg_start")
	return int(idx), int(term), lead != 0
}

// Kill, raft.go:377-385.  The harness has already taken the persister copy (config.go:110-126).
func (rf *Raft) Kill() {
	if atomic.CompareAndSwapInt32(&rf.dead, 0, 1) {
		rf.ps.SaveRaftState(rf.export())
		status(C.rg_crash(rf.grp.e, rf.grp.g, C.int32_t(rf.me)), "rg_crash")
	}
}

func (rf *Raft) killed() bool { return atomic.LoadInt32(&rf.dead) == 1 }

// export reads the peer's persisted state (persist(), raft.go:107-123) as the engine's blob.
func (rf *Raft) export() []byte {
	var size C.int64_t
	status(C.rg_snapshot_export(rf.grp.e, rf.grp.g, C.int32_t(rf.me), nil, 0, &size), "rg_snapshot_export")
	if size == 0 {
		return nil
	}
	buf := make([]byte, int(size))
	status(C.rg_snapshot_export(rf.grp.e, rf.grp.g, C.int32_t(rf.me), unsafe.Pointer(&buf[0]), size, &size), "rg_snapshot_export")
	return buf
}

// Make, raft.go:700-731.  peers is kept for len(peers) only: the peers of a group talk inside the GPU.
func Make(peers []*labrpc.ClientEnd, me int, persister *Persister, applyCh chan ApplyMsg) *Raft {
	grp := current
	rf := &Raft{grp: grp, me: me, applyCh: applyCh, ps: persister}
	blob := persister.ReadRaftState()
	if len(blob) > 0 || grp.booted[me] {
		var p unsafe.Pointer
		if len(blob) > 0 {
			p = unsafe.Pointer(&blob[0])
		}
		status(C.rg_snapshot_import(grp.e, grp.g, C.int32_t(me), p, C.int64_t(len(blob))), "rg_snapshot_import")
		status(C.rg_restart(grp.e, grp.g, C.int32_t(me)), "rg_restart") // Make + readPersist, raft.go:708-725
	}
	grp.mu.Lock()
	grp.booted[me] = true
	grp.rafts[me] = rf
	grp.mu.Unlock()
	return rf
}

// drive is the one long-running goroutine of a group: ticks of 10 ms, the persister kept current
// (persist() runs inside the engine), ApplyMsgs delivered in order (raft.go:672-688).
func (grp *Group) drive() {
	buf := make([]C.rg_apply_msg, 1024)
	for atomic.LoadInt32(&grp.stop) == 0 {
		time.Sleep(C.RG_TICK_MS * time.Millisecond)
		status(C.rg_step(grp.e, 1), "rg_step")
		grp.mu.Lock()
		rafts := append([]*Raft(nil), grp.rafts...)
		grp.mu.Unlock()
		for i, rf := range rafts {
			if rf == nil || rf.killed() {
				continue
			}
			rf.ps.SaveRaftState(rf.export())
			var n C.int64_t
			status(C.rg_drain_applied(grp.e, grp.g, C.int32_t(i), &buf[0], C.int64_t(len(buf)), &n), "rg_drain_applied")
			for k := 0; k < int(n); k++ {
				grp.mu.Lock()
				cmd := grp.cmds[uint64(buf[k].command)]
				grp.mu.Unlock()
				rf.applyCh <- ApplyMsg{true, cmd, int(buf[k].command_index)}
			}
		}
	}
}
