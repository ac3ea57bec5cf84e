package raft

// The calls src/raft/config.go makes into the engine binding.  config.go itself is the reference's file and
// is not copied here; a maintainer adds ONE line at each place named below, nothing else changes, and
// test_test.go runs unmodified.

/*
#include "raftgpu.h"
*/
import "C"

import "sync"

var (
	hooksMu sync.Mutex
	groupOf = map[*config]*Group{} // config has no field to spare: the binding keeps its group by pointer
)

func grpOf(cfg *config) *Group {
	hooksMu.Lock()
	defer hooksMu.Unlock()
	return groupOf[cfg]
}

// make_config, right after `cfg.net = labrpc.MakeNetwork()` (config.go:70):  hookMakeConfig(cfg, n, unreliable)
// The peers that the loop at config.go:87-91 makes with start1 are the fresh ones BeginConfig prepared.
func hookMakeConfig(cfg *config, n int, unreliable bool) {
	grp := BeginConfig(n, unreliable)
	hooksMu.Lock()
	groupOf[cfg] = grp
	hooksMu.Unlock()
}

// crash1, first line (config.go:100):  hookCrash(cfg, i)
// persist() runs inside the engine; the Persister the harness is about to copy (config.go:110-126) gets the
// peer's current blob first.  Then the engine's crash1: off the net with its server deleted (what
// cfg.net.DeleteServer(i), config.go:104, does to the labrpc network); rf.Kill() at config.go:117 finds it dead.
func hookCrash(cfg *config, i int) {
	if rf := cfg.rafts[i]; rf != nil { // (start1 of a server that never ran calls crash1 too: nothing to take down)
		rf.SyncPersister()
		grp := grpOf(cfg)
		status(C.rg_crash(grp.en.e, grp.g, C.int32_t(i)), "rg_crash")
	}
}

// start1, right before `rf := Make(...)` (config.go:205):  hookStart(cfg)
// Make finds its group through this (several configs may be alive, e.g. in kvraft's tests).
func hookStart(cfg *config) { current = grpOf(cfg) }

// connect / disconnect, first line (config.go:236, 259):  hookConnect(cfg, i, true|false)
func hookConnect(cfg *config, i int, on bool) { grpOf(cfg).SetConnected(i, on) }

// setunreliable / setlongreordering (config.go:289, 297):  hookRates(cfg, unrel, longrel)
// with the other flag as the config last set it.
var ratesOf = map[*config][2]bool{}

func hookUnreliable(cfg *config, unrel bool) {
	hooksMu.Lock()
	r := ratesOf[cfg]
	r[0] = unrel
	ratesOf[cfg] = r
	hooksMu.Unlock()
	grpOf(cfg).setRates(r[0], r[1])
}
func hookLongReordering(cfg *config, longrel bool) {
	hooksMu.Lock()
	r := ratesOf[cfg]
	r[1] = longrel
	ratesOf[cfg] = r
	hooksMu.Unlock()
	grpOf(cfg).setRates(r[0], r[1])
}

// rpcTotal / bytesTotal (config.go:284-294):  return hookRPCTotal(cfg) / hookBytesTotal(cfg)
func hookRPCTotal(cfg *config) int     { return grpOf(cfg).RPCTotal() }
func hookBytesTotal(cfg *config) int64 { return grpOf(cfg).BytesTotal() }

// cleanup, first line (config.go:225):  hookCleanup(cfg)
func hookCleanup(cfg *config) {
	grp := grpOf(cfg)
	hooksMu.Lock()
	delete(groupOf, cfg)
	delete(ratesOf, cfg)
	hooksMu.Unlock()
	grp.EndConfig()
}
