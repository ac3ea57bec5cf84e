package raft

// Deterministic lockstep driver for the reference's own Raft code.
//
// Copy this file into src/raft/ of the reference (next to raft.go; it is in package raft so that it can build
// &Raft{} values and reach their fields) and run
//
//	LOCKSTEP_CASES=<repo>/tests/golden/lockstep_cases.json LOCKSTEP_OUT=/tmp/go_lockstep.json go test -run TestLockstep
//
// then compare with  python <repo>/tests/golden/diff_go_lockstep.py /tmp/go_lockstep.json .
//
// What runs here is the reference itself wherever it is callable: rf.RequestVote and rf.AppendEntries
// (raft.go:200, 270), rf.Start (:344), rf.GetState (:90), rf.persist / rf.readPersist with the real Persister and
// gob (:107, :129).  What is buried in goroutines with sleeps -- election (:536-669), monitor (:387-530), apply
// (:672-688) -- and the labrpc network (labrpc.go:216-307) is restated statement by statement under the schedule
// of docs/LOCKSTEP.md, with the same seeded streams as oracle/raft_oracle.c, so that the quantities written at
// the end are those of tests/golden/lockstep_digests.json.  The handlers stamp rf.timer with the wall clock
// (:223, :287); the driver replaces the stamp by the simulated time right after the call.

import (
	"bytes"
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"io/ioutil"
	"os"
	"sort"
	"testing"

	"../labgob"
	"../labrpc"
)

const (
	lsTickMs          = 10
	lsElectionPeriod  = 50
	lsLeaderExtra     = 10
	lsMonitorPoll     = 20
	lsHbIdle          = 100
	lsHbBacklog       = 50
	lsCandidateWindow = 1000
	lsTimeoutBase     = 500
	lsTimeoutSpan     = 600
	lsRPCWait         = 25
	lsDelaySpan       = 27

	flagElectionSafety     = 0x01
	flagLogMatching        = 0x02
	flagLeaderCompleteness = 0x04
	flagStateMachine       = 0x08
	flagApplyOrder         = 0x10
	flagRefPanic           = 0x20
	flagLogOverflow        = 0x40

	streamElection = 1
	streamNet      = 2
	streamCmd      = 3
	streamPart     = 4
	streamChurn    = 5
	streamSched    = 6 // scenario schedules (rg_set_schedule) are not among this driver's cases; listed to keep the numbering whole
)

var lsCounterNames = []string{"group_ticks", "rpc_sent", "rpc_delivered", "rpc_replied", "rv_rounds", "elections_won",
	"ae_rounds", "entries_started", "entries_replicated", "entries_committed", "entries_applied", "persists",
	"persist_entries", "timeouts", "ae_conflicts", "stepdowns", "ae_delivered", "rpc_fixed_bytes", "ae_entries_sent"}

type lsConfig struct {
	NGroups, NPeers, LogCap                         int
	GroupBase                                       int64
	Seed                                            uint64
	ReqDrop, ReplyDrop, ReplyDelay                  int
	StartPeriod, StartFirstTick, StartMax, StartMode int
	PartPeriod, PartJitter                          int
	ChurnPeriod, ChurnDisc, ChurnRestart, ChurnCrash int
	AccountBytes                                    int
}

type lsCase struct {
	Name   string
	Config map[string]int64
	Stops  []int
}

type lsPeer struct {
	rf        *Raft
	ps        *Persister
	timeout   int64 // locals of election(), raft.go:543-547
	restart   bool
	elNext    int64
	monAlive  bool // the monitor goroutine, raft.go:387-394
	first     bool
	monNext   int64
	connected bool
	side      int
	rngCtr    uint32
	roundSeq  uint32
	applyHash uint64
	snapSync  int // entries of the log known equal to the persisted prefix (counts what a copying persist writes)
}

type lsGroup struct {
	peers                                  []*lsPeer
	key                                    uint64
	gid                                    int64
	startSeq, partNext, partEpoch, recMax  int
	flags                                  uint32
	applyRec                               []uint64
}

type lsEngine struct {
	cfg    lsConfig
	now    int
	groups []*lsGroup
	ctr    map[string]uint64
}

func sm64(x uint64) uint64 {
	x += 0x9E3779B97F4A7C15
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9
	x = (x ^ (x >> 27)) * 0x94D049BB133111EB
	return x ^ (x >> 31)
}
func streamHash(key uint64, stream uint64, a, b uint32) uint64 {
	return sm64(key ^ (stream << 56) ^ (uint64(a) << 32) ^ uint64(b))
}
func cmdOf(e Entry) uint64 {
	if v, ok := e.Command.(uint64); ok {
		return v
	}
	return 0 // the sentinel "init" of raft.go:710-715
}

// nextRandom(), raft.go:532-534, on the peer's seeded stream
func (g *lsGroup) nextRandom(p int) int64 {
	pr := g.peers[p]
	h := streamHash(g.key, streamElection, uint32(p), pr.rngCtr)
	pr.rngCtr++
	return int64(uint32(h>>32)%lsTimeoutSpan) + lsTimeoutBase
}
func sleepSlack(key uint64, p int, now int64) int64 {
	h := uint32(key) ^ (uint32(p) * 0x9E3779B1) ^ (uint32(now) * 0x85EBCA6B)
	h ^= h >> 16
	h *= 0x7FEB352D
	h ^= h >> 15
	h *= 0x846CA68B
	h ^= h >> 16
	return int64(h >> 30)
}
func turnStart(key uint64, tick int, P int) int {
	h := uint32(tick)*0x9E3779B1 + uint32(key)
	h ^= h >> 15
	h *= 0x85EBCA6B
	h ^= h >> 13
	return int(((h >> 16) * uint32(P)) >> 16)
}

// persisted reports what the Persister holds, decoded the way readPersist does (raft.go:135-141).
func (pr *lsPeer) persisted() (has bool, term int32, vote int, logs []Entry) {
	data := pr.ps.ReadRaftState()
	if len(data) < 1 {
		return false, 0, 0, nil
	}
	d := labgob.NewDecoder(bytes.NewBuffer(data))
	if d.Decode(&term) != nil || d.Decode(&vote) != nil || d.Decode(&logs) != nil {
		panic("lockstep: persisted state does not decode")
	}
	return true, term, vote, logs
}

// notePersist does the bookkeeping of one persist() call the reference just made: the call counter and the number
// of entries a persist that copies only what changed would have written.
func (e *lsEngine) notePersist(pr *lsPeer) {
	n := pr.rf.commitIndex + 1
	if n > e.cfg.LogCap {
		n = e.cfg.LogCap
	}
	if n > pr.snapSync {
		e.ctr["persist_entries"] += uint64(n - pr.snapSync)
		pr.snapSync = n
	}
	e.ctr["persists"]++
}

// makePeer is Make(), raft.go:700-731, without its two goroutines, plus the head of election(), :538-547.
func (e *lsEngine) makePeer(g *lsGroup, p int, nowMs int64) {
	pr := g.peers[p]
	rf := &Raft{}
	rf.peers = make([]*labrpc.ClientEnd, e.cfg.NPeers)
	rf.persister = pr.ps
	rf.me = p
	rf.currentTerm = 0
	rf.voteFor = -1
	rf.logs = []Entry{{Command: "init", Term: 0}}
	rf.commitIndex = 0
	rf.lastApplied = 0
	rf.timer = nowMs
	rf.applyCh = make(chan ApplyMsg, 1)
	rf.state = 0
	rf.stop = false
	rf.cnt = 0
	rf.heartbeatTimer = nowMs
	pr.rf = rf
	data := pr.ps.ReadRaftState()
	if len(data) >= 1 {
		_, _, _, logs := pr.persisted()
		if len(logs) >= 1 {
			rf.readPersist(data) // raft.go:129-152, persists once more at :147 with commitIndex still 0
			if len(logs) > pr.snapSync {
				pr.snapSync = len(logs)
			}
			saved := rf.commitIndex
			rf.commitIndex = 0
			e.notePersist(pr)
			rf.commitIndex = saved
		}
	}
	pr.monAlive, pr.first, pr.monNext = false, false, 0
	pr.restart = false
	pr.timeout = 0
	if !(rf.currentTerm == 0 && p == 0) { // raft.go:545-547
		pr.timeout = g.nextRandom(p)
	}
	pr.elNext = nowMs
}

func isLeader(rf *Raft) bool { _, l := rf.GetState(); return l }

// ---- network, labrpc.go:216-307 under docs/LOCKSTEP.md ----
const (
	netLost      = 0
	netReplyLost = 1
	netOK        = 2
)

func permilleOf(field16 uint32) int { return int((field16 * 1000) >> 16) }

func (e *lsEngine) netOutcome(g *lsGroup, s, j int, seq uint32) (out int, cost int64, wire bool) {
	a, b := g.peers[s], g.peers[j]
	cost = lsRPCWait
	// (every dead peer of this driver died by crash1, whose DeleteServer takes it off the net, config.go:99-127; a
	// peer that was only Kill()ed would still be served -- that verb is not among the driver's cases)
	if !(a.connected && b.connected && a.side == b.side) || b.rf.killed() {
		h := streamHash(g.key, streamNet, (uint32(s)<<8)|uint32(j), seq)
		ms := int64(((h >> 32) * 7000) >> 32) // labrpc.go:296: Call returns false after 0..6999 ms
		if ms < cost {
			cost = ms
		}
		return netLost, cost, false
	}
	c := &e.cfg
	if c.ReqDrop == 0 && c.ReplyDrop == 0 && c.ReplyDelay == 0 {
		return netOK, 0, true
	}
	h := streamHash(g.key, streamNet, (uint32(s)<<8)|uint32(j), seq)
	d := int64((((h >> 48) & 0xFF) * lsDelaySpan) >> 8)
	back := d
	if back > lsRPCWait {
		back = lsRPCWait
	}
	if permilleOf(uint32(h&0xFFFF)) < c.ReqDrop {
		return netLost, back, false
	}
	wire = !(permilleOf(uint32((h>>16)&0xFFFF)) < c.ReplyDrop)
	if !wire {
		return netReplyLost, back, false
	}
	if d >= lsRPCWait {
		return netReplyLost, cost, true
	}
	if permilleOf(uint32((h>>32)&0xFFFF)) < c.ReplyDelay {
		return netReplyLost, cost, true
	}
	return netOK, d, true
}

// callRequestVote runs the reference's handler on the target and does the driver's bookkeeping around it.
func (e *lsEngine) callRequestVote(g *lsGroup, to int, nowMs int64, args *RequestVoteArgs, reply *RequestVoteReply) {
	pr := g.peers[to]
	rf := pr.rf
	if rf.killed() {
		return
	}
	termBefore, timerBefore := rf.currentTerm, rf.timer
	rf.RequestVote(args, reply)
	if rf.currentTerm != termBefore { // persist() at raft.go:212
		e.notePersist(pr)
	}
	if rf.timer != timerBefore { // granted: persist() at :222, timer stamped at :223
		e.notePersist(pr)
		rf.timer = nowMs
	}
}

func (e *lsEngine) callAppendEntries(g *lsGroup, to int, nowMs int64, args *AppendEntriesArgs, reply *AppendEntriesReply) {
	pr := g.peers[to]
	rf := pr.rf
	if rf.killed() {
		return
	}
	termBefore := rf.currentTerm
	stale := args.Term < rf.currentTerm
	func() {
		defer func() {
			if r := recover(); r != nil { // the reference indexes past len(logs) in states the model flags
				g.flags |= flagRefPanic
				rf.mu.Unlock() // the handler panicked between Lock and Unlock (raft.go:274-318)
			}
		}()
		rf.AppendEntries(args, reply)
	}()
	if stale {
		return
	}
	rf.timer = nowMs // raft.go:287
	if rf.currentTerm != termBefore {
		e.notePersist(pr) // :284 (commitIndex is still the old one there; snapSync cannot grow from it)
	}
	if reply.Success {
		base := args.PrevLogIndex + 1
		if len(args.Entries) > 0 && base < pr.snapSync {
			pr.snapSync = base
		}
		if rf.commitIndex > len(rf.logs)-1 {
			g.flags |= flagRefPanic
		}
		e.notePersist(pr) // :300
		e.ctr["entries_replicated"] += uint64(len(args.Entries))
	} else {
		e.ctr["ae_conflicts"]++
	}
}

// ---- election loop iteration, raft.go:548-668 ----
func (e *lsEngine) electionStep(g *lsGroup, me int, nowMs int64) {
	pr := g.peers[me]
	rf := pr.rf
	l := e.cfg.NPeers
	state := rf.state
	timer := rf.timer
	switch state {
	case 0:
		if nowMs-timer > pr.timeout { // :557
			rf.state = 1
			pr.timeout = g.nextRandom(me)
			e.ctr["timeouts"]++
		}
	case 1:
		pr.timeout = lsCandidateWindow
		rf.timer = nowMs
		if !pr.restart {
			rf.currentTerm = rf.currentTerm + 1
		} else {
			pr.restart = false
		}
		rf.voteFor = me
		rf.persist()
		e.notePersist(pr)
		args := &RequestVoteArgs{Term: rf.currentTerm, CandidateId: me, LastLogIndex: rf.commitIndex, LastLogTerm: rf.logs[rf.commitIndex].Term}
		reply := &RequestVoteReply{Term: rf.currentTerm, VoteGranted: false}
		rf.persist()
		e.notePersist(pr)
		e.ctr["rv_rounds"]++
		seq := pr.roundSeq
		pr.roundSeq++
		voteCnt := 1
		var busy int64
		for i := 0; i < l && state == 1; i++ {
			state = rf.state
			if i != me {
				out, cost, wire := e.netOutcome(g, me, i, seq)
				busy += cost
				e.ctr["rpc_sent"]++
				if e.cfg.AccountBytes != 0 {
					e.ctr["rpc_fixed_bytes"] += 16
					if wire {
						e.ctr["rpc_fixed_bytes"] += 8
					}
				}
				ok := false
				if out != netLost {
					var w RequestVoteReply // what travels; gob leaves zero fields of the shared reply alone (labgob.go:116-121)
					e.callRequestVote(g, i, nowMs, args, &w)
					e.ctr["rpc_delivered"]++
					if out == netOK {
						ok = true
						if w.Term != 0 {
							reply.Term = w.Term
						}
						if w.VoteGranted {
							reply.VoteGranted = true
						}
						e.ctr["rpc_replied"]++
					}
				}
				if ok {
					if reply.VoteGranted {
						voteCnt++
						reply.VoteGranted = false
					}
					if rf.currentTerm != args.Term {
						break
					}
					if reply.Term > rf.currentTerm { // :622-627
						rf.state = 0
						rf.voteFor = -1
						rf.persist()
						e.notePersist(pr)
						e.ctr["stepdowns"]++
						break
					}
				}
			}
		}
		state = rf.state
		if !rf.killed() {
			switch state {
			case 0:
				pr.timeout = g.nextRandom(me) // :639
			case 1:
				if nowMs+busy-rf.timer < pr.timeout && 2*voteCnt > l { // :641
					rf.stop = false
					rf.cnt = 0
					rf.state = 2
					rf.heartbeatTimer = nowMs + busy
					rf.nextIndex = []int{}
					logLen := len(rf.logs)
					for i := 0; i < l; i++ {
						rf.nextIndex = append(rf.nextIndex, logLen)
					}
					rf.matchIndex = []int{}
					for i := 0; i < l; i++ {
						rf.matchIndex = append(rf.matchIndex, 0)
					}
					pr.monAlive, pr.first, pr.monNext = true, true, nowMs+busy // go monitor(rf)
					e.ctr["elections_won"]++
				} else {
					pr.restart = true
				}
			}
		}
		pr.elNext = nowMs + busy + lsElectionPeriod + sleepSlack(g.key, me, nowMs)
		return
	}
	extra := int64(0)
	if state == 2 {
		extra = lsLeaderExtra
	}
	pr.elNext = nowMs + lsElectionPeriod + extra + sleepSlack(g.key, me, nowMs)
}

// ---- leader loop, raft.go:395-529, until it would sleep or exit ----
func (e *lsEngine) monitorStep(g *lsGroup, me int, nowMs int64) {
	pr := g.peers[me]
	rf := pr.rf
	l := e.cfg.NPeers
	for {
		if rf.killed() || !isLeader(rf) { // :526-528
			pr.monAlive, pr.monNext = false, 0
			return
		}
		active := 1
		newCommitIndex := len(rf.logs) - 1 // :401-416
		for ; rf.state == 2 && newCommitIndex > rf.commitIndex && rf.logs[newCommitIndex].Term == rf.currentTerm; newCommitIndex-- {
			cnt := 1
			for i := 0; i < l; i++ {
				if i != me && rf.matchIndex[i] >= newCommitIndex {
					cnt++
				}
			}
			if 2*cnt > l && rf.logs[newCommitIndex].Term == rf.currentTerm {
				e.ctr["entries_committed"] += uint64(newCommitIndex - rf.commitIndex)
				rf.commitIndex = newCommitIndex
				pr.first = true
				rf.persist()
				e.notePersist(pr)
				break
			}
		}
		timer := rf.heartbeatTimer
		logLen := len(rf.logs)
		commitIndex := rf.commitIndex
		since := nowMs - timer
		if pr.first || since >= lsHbIdle || (commitIndex < logLen-1 && since >= lsHbBacklog) { // :421
			pr.first = false
			e.ctr["ae_rounds"]++
			seq := pr.roundSeq
			pr.roundSeq++
			var busy int64
			for i := 0; i < l; i++ {
				if !isLeader(rf) { // :424-427
					break
				}
				if i == me {
					continue
				}
				var args *AppendEntriesArgs
				if rf.nextIndex[i] <= logLen {
					args = &AppendEntriesArgs{Term: rf.currentTerm, LeaderId: me, PrevLogIndex: rf.nextIndex[i] - 1,
						PrevLogTerm: rf.logs[rf.nextIndex[i]-1].Term, Entries: rf.logs[rf.nextIndex[i]:logLen], LeaderCommit: rf.commitIndex}
				} else {
					break // :451-454
				}
				reply := &AppendEntriesReply{}
				out, cost, wire := e.netOutcome(g, me, i, seq)
				busy += cost
				e.ctr["rpc_sent"]++
				if e.cfg.AccountBytes != 0 {
					e.ctr["rpc_fixed_bytes"] += 24
					if wire {
						e.ctr["rpc_fixed_bytes"] += 16
					}
					e.ctr["ae_entries_sent"] += uint64(len(args.Entries))
				}
				ok := false
				if out != netLost {
					entries := append([]Entry(nil), args.Entries...) // the wire copies them (labrpc.go:91-94)
					args.Entries = entries
					e.callAppendEntries(g, i, nowMs, args, reply)
					e.ctr["rpc_delivered"]++
					e.ctr["ae_delivered"]++
					if out == netOK {
						ok = true
						e.ctr["rpc_replied"]++
					}
				}
				if ok {
					active++
					if rf.currentTerm != args.Term {
						break
					} else if reply.Term > rf.currentTerm { // :482-487
						rf.state = 0
						rf.currentTerm = reply.Term
						rf.persist()
						e.notePersist(pr)
						e.ctr["stepdowns"]++
						break
					} else {
						if reply.Success {
							rf.nextIndex[i] = logLen
							rf.matchIndex[i] = rf.nextIndex[i] - 1
						} else {
							if reply.ConflictTerm >= 0 {
								index := rf.nextIndex[i] - 1
								for rf.logs[index].Term > reply.ConflictTerm {
									index--
								}
								if rf.logs[index].Term == reply.ConflictTerm {
									rf.nextIndex[i] = index + 1
								} else {
									rf.nextIndex[i] = reply.ConflictFirstIndex
								}
							} else {
								rf.nextIndex[i] = reply.ConflictFirstIndex
							}
						}
					}
				}
			}
			if active == 1 && rf.cnt <= 3 { // :514-520
				rf.cnt++
				if rf.cnt > 3 {
					rf.stop = true
				}
			}
			rf.heartbeatTimer = nowMs + busy // :521
			if busy > 0 {
				pr.monNext = nowMs + busy
				return
			}
		} else {
			pr.monNext = nowMs + lsMonitorPoll // :524
			return
		}
	}
}

// ---- apply loop iteration, raft.go:676-686, and the tester's reader, config.go:170-194 ----
func (e *lsEngine) applyStep(g *lsGroup, p int) {
	pr := g.peers[p]
	rf := pr.rf
	for rf.commitIndex > rf.lastApplied {
		rf.lastApplied++
		idx := rf.lastApplied
		if idx >= len(rf.logs) {
			g.flags |= flagRefPanic
			if idx >= cap(rf.logs) {
				break // the reference has panicked; nothing more to restate
			}
		}
		cmd := cmdOf(rf.logs[:cap(rf.logs)][idx])
		wd := sm64(cmd ^ (uint64(uint32(rf.lastApplied)) << 40))
		lo := uint32(pr.applyHash) + uint32(wd)
		hi := uint32(pr.applyHash>>32) + uint32(wd>>32)
		pr.applyHash = uint64(hi)<<32 | uint64(lo)
		if rf.lastApplied <= g.recMax {
			if g.applyRec[idx] != cmd {
				g.flags |= flagStateMachine
			}
		} else {
			if rf.lastApplied != g.recMax+1 {
				g.flags |= flagApplyOrder
			}
			g.applyRec[idx] = cmd
			g.recMax = rf.lastApplied
		}
		e.ctr["entries_applied"]++
	}
}

func (e *lsEngine) startCmd(g *lsGroup, p int, cmd uint64) bool {
	pr := g.peers[p]
	rf := pr.rf
	if !isLeader(rf) {
		return false
	}
	if len(rf.logs) >= e.cfg.LogCap {
		g.flags |= flagLogOverflow
		return false
	}
	idx, _, ok := rf.Start(cmd) // raft.go:344-364, persists at :358
	if ok {
		if idx < pr.snapSync {
			pr.snapSync = idx
		}
		e.notePersist(pr)
		e.ctr["entries_started"]++
	}
	return ok
}

func (e *lsEngine) crash1(g *lsGroup, p int) { // config.go:99-127
	pr := g.peers[p]
	pr.connected = false
	pr.ps = pr.ps.Copy()
	pr.rf.Kill()
	np := MakePersister()
	np.SaveRaftState(pr.ps.ReadRaftState())
	pr.ps = np
}
func (e *lsEngine) start1(g *lsGroup, p int, nowMs int64) { // config.go:136-215
	e.crash1(g, p)
	pr := g.peers[p]
	pr.ps = pr.ps.Copy()
	// the lockstep model's bookkeeping of what the new instance's log shares with the persisted prefix
	e.makePeer(g, p, nowMs)
}

func (e *lsEngine) controlStep(g *lsGroup, tick int) {
	c := &e.cfg
	P := c.NPeers
	if c.PartPeriod > 0 && tick == g.partNext { // kvraft/test_test.go:124-143
		h := streamHash(g.key, streamPart, uint32(g.partEpoch), 0)
		for p := 0; p < P; p++ {
			g.peers[p].side = int((h >> uint(p)) & 1)
		}
		g.partNext = tick + c.PartPeriod + int(((h>>32)*uint64(uint32(c.PartJitter)))>>32)
		g.partEpoch++
	}
	if c.ChurnPeriod > 0 && tick > 0 && tick%c.ChurnPeriod == 0 { // test_test.go:893-919
		it := uint32(tick / c.ChurnPeriod)
		h1 := streamHash(g.key, streamChurn, it, 0)
		h2 := streamHash(g.key, streamChurn, it, 1)
		h3 := streamHash(g.key, streamChurn, it, 2)
		if int((h1&0xFFFFF)%1000) < c.ChurnDisc {
			g.peers[int((h1>>32)%uint64(P))].connected = false
		}
		if int((h2&0xFFFFF)%1000) < c.ChurnRestart {
			i := int((h2 >> 32) % uint64(P))
			if g.peers[i].rf.killed() {
				e.start1(g, i, int64(tick)*lsTickMs)
			}
			g.peers[i].connected = true
		}
		if int((h3&0xFFFFF)%1000) < c.ChurnCrash {
			i := int((h3 >> 32) % uint64(P))
			if !g.peers[i].rf.killed() {
				e.crash1(g, i)
			}
		}
	}
	if c.StartPeriod > 0 && tick >= c.StartFirstTick && (tick-c.StartFirstTick)%c.StartPeriod == 0 { // config.go:436-451 / test_test.go:697-703
		for p := 0; p < P; p++ {
			if g.startSeq >= c.StartMax {
				break
			}
			pr := g.peers[p]
			if pr.rf.killed() || (c.StartMode == 0 && !pr.connected) || !isLeader(pr.rf) {
				continue
			}
			if e.startCmd(g, p, streamHash(g.key, streamCmd, 0, uint32(g.startSeq))) {
				g.startSeq++
				if c.StartMode == 0 {
					break
				}
			}
		}
	}
}

func (e *lsEngine) checkLeaders(g *lsGroup) { // config.go:307-320
	for a := 0; a < e.cfg.NPeers; a++ {
		x := g.peers[a]
		if !x.connected || !isLeader(x.rf) {
			continue
		}
		for b := a + 1; b < e.cfg.NPeers; b++ {
			y := g.peers[b]
			if y.connected && isLeader(y.rf) && y.rf.currentTerm == x.rf.currentTerm {
				g.flags |= flagElectionSafety
			}
		}
	}
}

func (e *lsEngine) tickGroup(g *lsGroup, tick int) {
	nowMs := int64(tick) * lsTickMs
	e.controlStep(g, tick)
	P := e.cfg.NPeers
	start := turnStart(g.key, tick, P)
	for q := 0; q < P; q++ {
		p := (start + q) % P
		pr := g.peers[p]
		if pr.rf.killed() {
			continue
		}
		if pr.elNext <= nowMs {
			e.electionStep(g, p, nowMs)
		}
		if !pr.rf.killed() && pr.monAlive && pr.monNext <= nowMs {
			e.monitorStep(g, p, nowMs)
		}
	}
	for p := 0; p < P; p++ {
		if !g.peers[p].rf.killed() {
			e.applyStep(g, p)
		}
	}
	e.checkLeaders(g)
}

func newEngine(cfg lsConfig) *lsEngine {
	e := &lsEngine{cfg: cfg, ctr: map[string]uint64{}}
	for _, n := range lsCounterNames {
		e.ctr[n] = 0
	}
	for gi := 0; gi < cfg.NGroups; gi++ {
		g := &lsGroup{gid: cfg.GroupBase + int64(gi), applyRec: make([]uint64, cfg.LogCap)}
		g.key = sm64(sm64(cfg.Seed) ^ uint64(g.gid))
		if cfg.PartPeriod > 0 {
			g.partNext = cfg.PartPeriod
		}
		for p := 0; p < cfg.NPeers; p++ {
			g.peers = append(g.peers, &lsPeer{ps: MakePersister(), connected: true, snapSync: 1})
		}
		boot := map[string]uint64{}
		for k, v := range e.ctr {
			boot[k] = v
		}
		for p := 0; p < cfg.NPeers; p++ {
			e.makePeer(g, p, 0)
		}
		e.ctr = boot // Make at tick 0 is not counted
		e.groups = append(e.groups, g)
	}
	return e
}

func (e *lsEngine) step(n int) {
	for _, g := range e.groups {
		for t := 0; t < n; t++ {
			e.tickGroup(g, e.now+t)
		}
	}
	e.ctr["group_ticks"] += uint64(len(e.groups)) * uint64(n)
	e.now += n
}

// ---- safety scan and digests: the quantities of tests/golden/lockstep_digests.json ----
func (e *lsEngine) safetyCheck() (uint32, uint32) {
	for _, g := range e.groups {
		P := e.cfg.NPeers
		for a := 0; a < P; a++ {
			for b := a + 1; b < P; b++ {
				x, y := g.peers[a].rf, g.peers[b].rf
				n := len(x.logs)
				if len(y.logs) < n {
					n = len(y.logs)
				}
				for i := 1; i < n; i++ {
					if x.logs[i].Term == y.logs[i].Term && (cmdOf(x.logs[i]) != cmdOf(y.logs[i]) || x.logs[i-1].Term != y.logs[i-1].Term) {
						g.flags |= flagLogMatching
					}
				}
			}
		}
		for a := 0; a < P; a++ {
			x := g.peers[a].rf
			if !isLeader(x) {
				continue
			}
			for i := 1; i <= g.recMax; i++ {
				if i >= len(x.logs) || cmdOf(x.logs[i]) != g.applyRec[i] {
					g.flags |= flagLeaderCompleteness
				}
			}
		}
	}
	var or, n uint32
	for _, g := range e.groups {
		or |= g.flags
		if g.flags != 0 {
			n++
		}
	}
	return or, n
}

func elemHash(peer, field, idx uint32, v uint64) uint64 {
	return sm64(sm64((uint64(peer)<<48)^(uint64(field)<<32)^uint64(idx)) ^ v)
}
func b2i(b bool) int32 {
	if b {
		return 1
	}
	return 0
}

// peerWords returns the 25 per-peer words in the order of the RG_F_* field ids of include/raftgpu.h.
func (pr *lsPeer) peerWords() []int32 {
	rf := pr.rf
	has, st, sv, slog := pr.persisted()
	return []int32{rf.currentTerm, int32(rf.voteFor), rf.state, int32(rf.commitIndex), int32(rf.lastApplied), int32(len(rf.logs)),
		int32(rf.timer), int32(pr.timeout), int32(rf.heartbeatTimer), b2i(rf.stop), int32(rf.cnt), b2i(rf.killed()),
		b2i(pr.restart), b2i(pr.first), b2i(pr.monAlive), int32(pr.elNext), int32(pr.monNext), int32(pr.rngCtr), int32(pr.roundSeq),
		b2i(has), st, int32(sv), int32(len(slog)), b2i(pr.connected), int32(pr.side)}
}

func (e *lsEngine) stateHash() []uint64 {
	out := make([]uint64, len(e.groups))
	P := e.cfg.NPeers
	for gi, g := range e.groups {
		var s uint64
		for p, pr := range g.peers {
			rf := pr.rf
			for f, v := range pr.peerWords() {
				s += elemHash(uint32(p), uint32(f), 0, uint64(uint32(v)))
			}
			s += elemHash(uint32(p), 32, 0, pr.applyHash)
			for i := 0; i < P; i++ {
				var nx, mx int
				if i < len(rf.nextIndex) {
					nx, mx = rf.nextIndex[i], rf.matchIndex[i]
				}
				s += elemHash(uint32(p), 40, uint32(i), uint64(uint32(int32(nx))))
				s += elemHash(uint32(p), 41, uint32(i), uint64(uint32(int32(mx))))
			}
			for i := 0; i < len(rf.logs) && i < e.cfg.LogCap; i++ {
				s += elemHash(uint32(p), 48, uint32(i), uint64(uint32(rf.logs[i].Term)))
				s += elemHash(uint32(p), 56, uint32(i), cmdOf(rf.logs[i]))
			}
			_, _, _, slog := pr.persisted()
			for i := 0; i < len(slog) && i < e.cfg.LogCap; i++ {
				s += elemHash(uint32(p), 49, uint32(i), uint64(uint32(slog[i].Term)))
				s += elemHash(uint32(p), 57, uint32(i), cmdOf(slog[i]))
			}
		}
		for f, v := range []int32{int32(g.startSeq), int32(g.partNext), int32(g.partEpoch), int32(g.recMax), int32(g.flags)} {
			s += elemHash(255, uint32(64+f), 0, uint64(uint32(v)))
		}
		for i := 1; i <= g.recMax && i < e.cfg.LogCap; i++ {
			s += elemHash(255, 72, uint32(i), g.applyRec[i])
		}
		out[gi] = sm64(s ^ uint64(g.gid))
	}
	return out
}

func (e *lsEngine) snapshot(tick int) map[string]interface{} {
	or, n := e.safetyCheck()
	h := e.stateHash()
	buf := make([]byte, 8*len(h))
	for i, v := range h {
		binary.LittleEndian.PutUint64(buf[8*i:], v)
	}
	sum := sha256.Sum256(buf)
	out := map[string]interface{}{"tick": tick, "safety": []uint32{or, n}, "state_sha256": hex.EncodeToString(sum[:])}
	ctr := map[string]uint64{}
	for k, v := range e.ctr {
		ctr[k] = v
	}
	out["counters"] = ctr
	sums := map[string]uint64{}
	for _, g := range e.groups {
		for _, pr := range g.peers {
			w := pr.peerWords()
			for i, name := range []string{"term", "vote", "state", "commit", "applied", "loglen"} {
				sums[name] += uint64(int64(w[i]))
			}
			sums["applyhash"] += pr.applyHash
		}
		sums["recmax"] += uint64(int64(g.recMax))
		sums["gflags"] += uint64(g.flags)
	}
	for k, v := range sums {
		out["sum_"+k] = v
	}
	return out
}

func cfgFrom(m map[string]int64) lsConfig {
	get := func(k string, d int64) int64 {
		if v, ok := m[k]; ok {
			return v
		}
		return d
	}
	return lsConfig{NGroups: int(get("n_groups", 1)), NPeers: int(get("n_peers", 3)), LogCap: int(get("log_cap", 64)),
		GroupBase: get("group_base", 0), Seed: uint64(get("seed", 1)),
		ReqDrop: int(get("req_drop_permille", 0)), ReplyDrop: int(get("reply_drop_permille", 0)), ReplyDelay: int(get("reply_delay_permille", 0)),
		StartPeriod: int(get("start_period", 0)), StartFirstTick: int(get("start_first_tick", 0)), StartMax: int(get("start_max", 0)),
		StartMode: int(get("start_mode", 0)), PartPeriod: int(get("part_period", 0)), PartJitter: int(get("part_jitter", 0)),
		ChurnPeriod: int(get("churn_period", 0)), ChurnDisc: int(get("churn_disconnect_permille", 0)),
		ChurnRestart: int(get("churn_restart_permille", 0)), ChurnCrash: int(get("churn_crash_permille", 0)),
		AccountBytes: int(get("account_bytes", 0))}
}

func TestLockstep(t *testing.T) {
	path := os.Getenv("LOCKSTEP_CASES")
	if path == "" {
		t.Skip("set LOCKSTEP_CASES to tests/golden/lockstep_cases.json of the engine repository")
	}
	raw, err := ioutil.ReadFile(path)
	if err != nil {
		t.Fatal(err)
	}
	var cases []lsCase
	if err := json.Unmarshal(raw, &cases); err != nil {
		t.Fatal(err)
	}
	sort.Slice(cases, func(i, j int) bool { return cases[i].Name < cases[j].Name })
	result := map[string][]map[string]interface{}{}
	for _, c := range cases {
		e := newEngine(cfgFrom(c.Config))
		done := 0
		for _, stop := range c.Stops {
			e.step(stop - done)
			done = stop
			result[c.Name] = append(result[c.Name], e.snapshot(stop))
		}
	}
	out, _ := json.MarshalIndent(result, "", " ")
	if dst := os.Getenv("LOCKSTEP_OUT"); dst != "" {
		if err := ioutil.WriteFile(dst, out, 0644); err != nil {
			t.Fatal(err)
		}
	} else {
		os.Stdout.Write(out)
	}
}
