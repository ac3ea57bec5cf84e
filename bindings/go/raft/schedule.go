package raft

// Scenario schedules and the per-configuration kernel from Go: the two engine calls that have no counterpart in
// the reference's package API.  A schedule is the tester's verbs (config.go:99-278, 302-332, 430-477) as a
// tick-sorted table that EVERY group of an engine executes on its own peers inside the tick kernel
// (rg_set_schedule); tests/schedules.py in this repository writes eleven of test_test.go's tests that way.

/*
#include "raftgpu.h"
*/
import "C"

// SchedOp is one row of the table (rg_sched_op): at Tick, Op on selector/register A with argument B.
type SchedOp struct {
	Tick, Op, A, B int32
}

// Ops and selectors, as in include/raftgpu.h.
const (
	OpConnect        = int32(C.RG_OP_CONNECT)
	OpDisconnect     = int32(C.RG_OP_DISCONNECT)
	OpCrash          = int32(C.RG_OP_CRASH)
	OpRestart        = int32(C.RG_OP_RESTART)
	OpStart          = int32(C.RG_OP_START)
	OpStartAny       = int32(C.RG_OP_START_ANY)
	OpStartAll       = int32(C.RG_OP_START_ALL)
	OpPickLeader     = int32(C.RG_OP_PICK_LEADER)
	OpPickRandom     = int32(C.RG_OP_PICK_RANDOM)
	OpPickOther      = int32(C.RG_OP_PICK_OTHER)
	OpRestartIfFewer = int32(C.RG_OP_RESTART_IF_FEWER)
	OpConnectIfFewer = int32(C.RG_OP_CONNECT_IF_FEWER)
	SelAll           = int32(C.RG_SEL_ALL)
)

// SelReg names the peer k places after the one register r holds (RG_SEL_REG).
func SelReg(r, k int32) int32 { return 0x100 | r<<4 | k }

// SetSchedule replaces the schedule of the engine that holds n-peer configs (nil: none).
func SetSchedule(n int, ops []SchedOp) {
	en := engineFor(n)
	if len(ops) == 0 {
		status(C.rg_set_schedule(en.e, nil, 0), "rg_set_schedule")
		return
	}
	cops := make([]C.rg_sched_op, len(ops))
	for i, o := range ops {
		cops[i] = C.rg_sched_op{tick: C.int32_t(o.Tick), op: C.int32_t(o.Op), a: C.int32_t(o.A), b: C.int32_t(o.B)}
	}
	status(C.rg_set_schedule(en.e, &cops[0], C.int64_t(len(cops))), "rg_set_schedule")
}

// Specialize compiles the tick kernel for the configuration of the engine that holds n-peer configs
// (rg_specialize): worth its second for long runs.  It reports whether the specialised kernel is in use.
func Specialize(n int) bool {
	return C.rg_specialize(engineFor(n).e) == C.RG_OK
}
