"""The reference's per-peer API over the batched engine.

Mirrors, name for name, what a service or tester uses of the reference
(reference: src/raft/raft.go:42-46 ApplyMsg, :700 Make, :344 Start, :90
GetState, :377 Kill; src/raft/persister.go:20-70 Persister):

    net  = Network(engine, group)          # stands where labrpc.Network stood
    rf   = Make(net.ends(i), i, persister, applyCh)
    rf.Start(cmd) -> (index, term, isLeader)
    rf.GetState() -> (term, isLeader)
    rf.Kill()

`net.ends(i)[j].Call("Raft.RequestVote" | "Raft.AppendEntries", args, reply)` is the labrpc-shaped way to
the exported handlers (reference: src/labrpc/labrpc.go:84-121, src/raft/raft.go:161-195).

Time does not pass by itself: `net.advance(ticks)` steps the engine (every
group it holds) and then delivers the ApplyMsgs each live peer emitted to its
applyCh, in index order, as the reference's apply goroutine does
(raft.go:672-688).  Commands are arbitrary Python values; the engine carries a
64-bit handle per command and the Network keeps the table.
"""
import struct


class ApplyMsg(object):
    """reference: src/raft/raft.go:42-46"""
    __slots__ = ("CommandValid", "Command", "CommandIndex")

    def __init__(self, CommandValid, Command, CommandIndex):
        self.CommandValid, self.Command, self.CommandIndex = CommandValid, Command, CommandIndex

    def __repr__(self):
        return "ApplyMsg(%r, %r, %r)" % (self.CommandValid, self.Command, self.CommandIndex)


class Persister(object):
    """reference: src/raft/persister.go:14-70.  The Raft state blob is the
    engine's snapshot-array export (header term, vote, length; then entries)."""

    def __init__(self):
        self.raftstate = b""
        self.snapshot = b""
        self._bound = None      # (network, peer) once handed to Make

    def Copy(self):
        np_ = Persister()
        np_.raftstate = self.ReadRaftState()
        np_.snapshot = self.snapshot
        return np_

    def SaveRaftState(self, state):
        self.raftstate = bytes(state)
        if self._bound:
            net, i = self._bound
            net.engine.snapshot_import(net.group, i, self.raftstate)

    def ReadRaftState(self):
        if self._bound:
            net, i = self._bound
            if net.persisters[i] is self:
                self.raftstate = net.engine.snapshot_export(net.group, i)
        return self.raftstate

    def RaftStateSize(self):
        return len(self.ReadRaftState())

    def SaveStateAndSnapshot(self, state, snapshot):
        self.SaveRaftState(state)
        self.snapshot = bytes(snapshot)

    def ReadSnapshot(self):
        return self.snapshot

    def SnapshotSize(self):
        return len(self.snapshot)


def MakePersister():
    return Persister()


def decode_raft_state(blob):
    """(term, vote, [(term, command_handle), ...]) of a Persister blob."""
    if not blob:
        return None
    term, vote, n, _ = struct.unpack_from("<iiii", blob, 0)
    ents = [struct.unpack_from("<iiQ", blob, 16 + 16 * i) for i in range(n)]
    return term, vote, [(t, c) for t, _, c in ents]


class RequestVoteArgs(object):
    """reference: src/raft/raft.go:161-168"""
    def __init__(self, Term=0, CandidateId=0, LastLogIndex=0, LastLogTerm=0):
        self.Term, self.CandidateId, self.LastLogIndex, self.LastLogTerm = Term, CandidateId, LastLogIndex, LastLogTerm


class RequestVoteReply(object):
    """reference: src/raft/raft.go:174-177"""
    def __init__(self, Term=0, VoteGranted=False):
        self.Term, self.VoteGranted = Term, VoteGranted


class AppendEntriesArgs(object):
    """reference: src/raft/raft.go:179-186; Entries = [(term, command), ...]"""
    def __init__(self, Term=0, LeaderId=0, PrevLogIndex=0, PrevLogTerm=0, Entries=(), LeaderCommit=0):
        self.Term, self.LeaderId, self.PrevLogIndex, self.PrevLogTerm = Term, LeaderId, PrevLogIndex, PrevLogTerm
        self.Entries, self.LeaderCommit = list(Entries), LeaderCommit


class AppendEntriesReply(object):
    """reference: src/raft/raft.go:188-195"""
    def __init__(self, Term=0, Success=False, ConflictTerm=0, ConflictFirstIndex=0):
        self.Term, self.Success, self.ConflictTerm, self.ConflictFirstIndex = Term, Success, ConflictTerm, ConflictFirstIndex


class ClientEnd(object):
    """The end server `me` uses to reach server `to` (reference: src/labrpc/labrpc.go:84-121).  Call returns
    False when the network loses the request -- an end of a disconnected server (config.go:235-278) or a killed
    target (labrpc.go:268-272).  A reply decodes the way gob does: zero fields of the wire value leave the
    caller's object alone (labgob.go:116-121), which the reference's election loop relies on (raft.go:585-616)."""

    def __init__(self, net, me, to):
        self.net, self.me, self.to = net, me, to

    def Call(self, svcMeth, args, reply):
        net = self.net
        target = net.rafts[self.to]
        if not (net.connected[self.me] and net.connected[self.to]) or target is None or target.killed():
            return False
        eng, g = net.engine, net.group
        if svcMeth == "Raft.RequestVote":
            term, granted = eng.request_vote(g, self.to, args.Term, args.CandidateId, args.LastLogIndex, args.LastLogTerm)
            if term != 0:
                reply.Term = term
            if granted:
                reply.VoteGranted = True
            return True
        if svcMeth == "Raft.AppendEntries":
            entries = [(t, net.handle_for(c)) for t, c in args.Entries]
            term, ok, ct, cf = eng.append_entries(g, self.to, args.Term, args.LeaderId, args.PrevLogIndex, args.PrevLogTerm,
                                                  entries, args.LeaderCommit)
            if term != 0:
                reply.Term = term
            if ok:
                reply.Success = True
            if ct != 0:
                reply.ConflictTerm = ct
            if cf != 0:
                reply.ConflictFirstIndex = cf
            return True
        return False                     # unknown service or method (labrpc.go:440-442)


class Network(object):
    """One Raft group of the engine, standing in for the labrpc.Network a
    tester builds per group (reference: src/raft/config.go:70).  On an engine that has already run, the group
    is first given fresh peers (rg_reset_groups): a new tester config starts from Make, not from whatever the
    group's previous peers left."""

    def __init__(self, engine, group=0, reset=None):
        self.engine, self.group, self.n = engine, group, engine.P
        if reset is None:
            reset = engine.now() > 0
        if reset:
            engine.reset_groups([group])
        self.connected = [True] * self.n
        self.rafts = [None] * self.n
        self.persisters = [None] * self.n
        self.apply_chs = [None] * self.n
        self._cmd_of = {0: "init"}     # handle -> command; entry 0 is the sentinel (raft.go:710-715)
        self._next_handle = 1
        self._booted = [False] * self.n

    def ends(self, i):
        return _Ends(self, i)

    def handle_for(self, command):
        h = self._next_handle
        self._next_handle += 1
        self._cmd_of[h] = command
        return h

    def command_of(self, handle):
        return self._cmd_of.get(handle, handle)

    def advance(self, ticks=1):
        self.engine.step(ticks)
        self.deliver()

    def deliver(self):
        for i, rf in enumerate(self.rafts):
            if rf is None or rf.killed():
                continue
            for idx, h in self.engine.drain_applied(self.group, i):
                ch = self.apply_chs[i]
                if ch is not None:
                    ch.put(ApplyMsg(True, self.command_of(h), idx))


class _Ends(object):
    """What Make receives as `peers`: this peer's view of the group."""

    def __init__(self, net, me):
        self.net, self.me = net, me

    def __len__(self):
        return self.net.n

    def __getitem__(self, j):
        return ClientEnd(self.net, self.me, j)


class Raft(object):
    def __init__(self, net, me):
        self._net, self.me, self._dead = net, me, False

    def Start(self, command):
        """reference: src/raft/raft.go:344-364"""
        if self._dead:
            t, _ = self._net.engine.get_state(self._net.group, self.me)
            return -1, t, False
        h = self._net.handle_for(command)
        return self._net.engine.start(self._net.group, self.me, h)

    def GetState(self):
        """reference: src/raft/raft.go:90-100"""
        t, l = self._net.engine.get_state(self._net.group, self.me)
        return t, (l and not self._dead)

    def Kill(self):
        """reference: src/raft/raft.go:377-380: the dead flag alone (rg_kill).  The peer stays on the net; taking it
        off is the tester's crash1 (Config.crash1 -> rg_crash)."""
        if not self._dead:
            self._dead = True
            self._net.engine.kill(self._net.group, self.me)

    def killed(self):
        return self._dead


def Make(peers, me, persister, applyCh):
    """reference: src/raft/raft.go:700-731.  `peers` is Network.ends(me).

    A peer that was never started keeps the engine's boot state when the
    persister is empty; otherwise the persister's blob becomes the peer's
    snapshot arrays and the peer is rebuilt from them (Make + readPersist).
    As in the harness (config.go:136-151), a re-made peer is off the network
    until it is connected again."""
    net = peers.net
    eng, g = net.engine, net.group
    old = net.rafts[me]
    if old is not None and not old.killed():
        old.Kill()
    blob = persister.ReadRaftState() if persister is not None else b""
    fresh = not net._booted[me] and not blob          # the peer rg_create / rg_reset_groups made is this one
    if persister is None:
        persister = Persister()
    persister._bound = (net, me)
    net.persisters[me] = persister
    if not fresh:
        eng.snapshot_import(g, me, blob)
        eng.restart(g, me)
    net._booted[me] = True
    rf = Raft(net, me)
    net.rafts[me] = rf
    net.apply_chs[me] = applyCh
    return rf
