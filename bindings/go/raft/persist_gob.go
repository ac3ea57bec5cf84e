package raft

// The Persister's raftstate in the reference's own encoding.  persist() (raft.go:116-122) writes
// labgob(currentTerm int32), labgob(voteFor int), labgob(logs[0:commitIndex+1] []Entry); readPersist()
// (raft.go:135-146) reads them back in that order.  The engine keeps the same three things as a flat record
// (rg_snapshot_export: int32 term, int32 vote, int32 len, int32 0, len x {int32 term, int32 0, uint64
// handle}), with commands as 64-bit handles into the config's command table.  These two functions convert
// between them, so that what the tester -- or any code written against the reference -- finds in
// Persister.ReadRaftState() decodes with the reference's own readPersist, and a blob the reference wrote
// restores a peer of the engine.

import (
	"bytes"
	"encoding/binary"
	"errors"

	"../labgob"
)

const recHeader, recEntry = 16, 16

// gobFromRecord renders an engine record as persist() would have written it.  Entry 0 is the sentinel
// Make() puts there (raft.go:710-715: Command "init", Term 0).
func (grp *Group) gobFromRecord(rec []byte) []byte {
	le := binary.LittleEndian
	term := int32(le.Uint32(rec[0:]))
	vote := int(int32(le.Uint32(rec[4:])))
	n := int(int32(le.Uint32(rec[8:])))
	logs := make([]Entry, n)
	grp.mu.Lock()
	for i := 0; i < n; i++ {
		o := recHeader + recEntry*i
		logs[i].Term = int32(le.Uint32(rec[o:]))
		if i == 0 {
			logs[i].Command = "init"
		} else {
			logs[i].Command = grp.cmds[le.Uint64(rec[o+8:])]
		}
	}
	grp.mu.Unlock()
	w := new(bytes.Buffer)
	e := labgob.NewEncoder(w)
	e.Encode(term)
	e.Encode(vote)
	e.Encode(logs)
	return w.Bytes()
}

// recordFromGob is readPersist's decode (raft.go:135-141) followed by the engine's record layout.  Commands
// get fresh handles in this config's table: a blob may come from another config's Persister
// (config.go:110-126 copies Persisters, never commands).  A blob that does not decode is an error, where the
// reference logs and starts empty (raft.go:141-142): the caller does the same.
func (grp *Group) recordFromGob(data []byte) ([]byte, error) {
	d := labgob.NewDecoder(bytes.NewBuffer(data))
	var term int32
	var vote int
	var logs []Entry
	if d.Decode(&term) != nil || d.Decode(&vote) != nil || d.Decode(&logs) != nil {
		return nil, errors.New("raftgpu: raftstate does not decode as (int32, int, []Entry)")
	}
	le := binary.LittleEndian
	rec := make([]byte, recHeader+recEntry*len(logs))
	le.PutUint32(rec[0:], uint32(term))
	le.PutUint32(rec[4:], uint32(int32(vote)))
	le.PutUint32(rec[8:], uint32(len(logs)))
	grp.mu.Lock()
	for i, en := range logs {
		o := recHeader + recEntry*i
		le.PutUint32(rec[o:], uint32(en.Term))
		if i > 0 {
			h := grp.next
			grp.next++
			grp.cmds[h] = en.Command
			le.PutUint64(rec[o+8:], h)
		}
	}
	grp.mu.Unlock()
	return rec, nil
}
