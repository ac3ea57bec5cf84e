package raft

// A labrpc-shaped end for the engine's exported handlers (labrpc.go:84-121: ClientEnd.Call(svcMeth, args,
// reply) bool).  Code that holds `peers []*labrpc.ClientEnd` and calls
// peers[i].Call("Raft.RequestVote" | "Raft.AppendEntries", &args, &reply) -- a test that injects RPCs, a
// peer living outside the engine -- uses EngineEnd the same way: the request is handled by peer `to` of the
// group (rg_rpc_request_vote / rg_rpc_append_entries, the reference's handlers raft.go:200-232, 270-321).

/*
#include "raftgpu.h"
*/
import "C"

import "unsafe"

// EngineEnd is the end server `from` would use to reach server `to` (config.go:147-150).
type EngineEnd struct {
	grp      *Group
	from, to int
}

func (grp *Group) MakeEnd(from, to int) *EngineEnd { return &EngineEnd{grp: grp, from: from, to: to} }

// Call returns false when the network loses the request: an end of a disconnected server (config.go:235-278)
// or a killed target (labrpc.go:268-272).  A reply decodes the way gob does: zero fields of the wire value
// leave the caller's struct alone (labgob.go:116-121), which the reference's election loop relies on
// (raft.go:585-616).
func (end *EngineEnd) Call(svcMeth string, args interface{}, reply interface{}) bool {
	grp := end.grp
	grp.mu.Lock()
	up := grp.connected[end.from] && grp.connected[end.to]
	target := grp.rafts[end.to]
	grp.mu.Unlock()
	if !up || target == nil || target.killed() {
		return false
	}
	g, p := grp.g, C.int32_t(end.to)
	switch svcMeth {
	case "Raft.RequestVote":
		a, r := args.(*RequestVoteArgs), reply.(*RequestVoteReply)
		ca := C.rg_vote_args{term: C.int32_t(a.Term), candidate: C.int32_t(a.CandidateId),
			last_log_index: C.int32_t(a.LastLogIndex), last_log_term: C.int32_t(a.LastLogTerm)}
		var cr C.rg_vote_reply
		status(C.rg_rpc_request_vote(grp.en.e, 1, &g, &p, &ca, &cr), "rg_rpc_request_vote")
		if cr.term != 0 {
			r.Term = int32(cr.term)
		}
		if cr.granted != 0 {
			r.VoteGranted = true
		}
		return true
	case "Raft.AppendEntries":
		a, r := args.(*AppendEntriesArgs), reply.(*AppendEntriesReply)
		n := len(a.Entries)
		terms, cmds := make([]C.int32_t, n+1), make([]C.uint64_t, n+1)
		grp.mu.Lock()
		for i, en := range a.Entries { // commands travel as handles of the group's table
			h := grp.next
			grp.next++
			grp.cmds[h] = en.Command
			terms[i], cmds[i] = C.int32_t(en.Term), C.uint64_t(h)
		}
		grp.mu.Unlock()
		ca := C.rg_append_args{term: C.int32_t(a.Term), leader: C.int32_t(a.LeaderId),
			prev_log_index: C.int32_t(a.PrevLogIndex), prev_log_term: C.int32_t(a.PrevLogTerm),
			leader_commit: C.int32_t(a.LeaderCommit), n_entries: C.int32_t(n), entry_off: 0}
		var cr C.rg_append_reply
		status(C.rg_rpc_append_entries(grp.en.e, 1, &g, &p, &ca, (*C.int32_t)(unsafe.Pointer(&terms[0])),
			(*C.uint64_t)(unsafe.Pointer(&cmds[0])), C.int64_t(n), &cr), "rg_rpc_append_entries")
		if cr.term != 0 {
			r.Term = int32(cr.term)
		}
		if cr.success != 0 {
			r.Success = true
		}
		if cr.conflict_term != 0 {
			r.ConflictTerm = int32(cr.conflict_term)
		}
		if cr.conflict_first_index != 0 {
			r.ConflictFirstIndex = int(cr.conflict_first_index)
		}
		return true
	}
	return false // labrpc: unknown service or method (labrpc.go:440-442 logs and fails)
}
