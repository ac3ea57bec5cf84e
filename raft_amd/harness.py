"""Tester harness in tick units: the verbs of the reference's config.go over
one group of the engine, so its test scenarios can be written the way the
reference writes them (reference: src/raft/config.go:60-509).

Sleeps become `advance(ticks)` (10 ms per tick).  The apply-channel reader
(config.go:168-203) is `_reader`: it records every ApplyMsg per server and
checks that no two servers applied different commands at one index and that
no server skipped an index.
"""
import queue

from .raft import Make, MakePersister, Network

ELECTION_TIMEOUT_TICKS = 100      # RaftElectionTimeout = 1000 ms (test_test.go:20)


class TestFailure(AssertionError):
    __test__ = False     # not a pytest class


def park_all(engine):
    """An engine that serves one tester config per group (the Go binding's engineFor): rg_create boots every group,
    and a group no config has taken must not elect, heartbeat or count RPCs -- the configs read engine-wide counters
    (config.go:284-294).  Every peer crashed and off the net, until a Config's rg_reset_groups makes its group's
    peers afresh."""
    import numpy as np
    engine.write("connected", np.zeros((engine.G, engine.P), dtype=np.int32))
    engine.write("dead", np.ones((engine.G, engine.P), dtype=np.int32))


class Config(object):
    """make_config (config.go:60-96) for group `group` of `engine`."""

    def __init__(self, engine, group=0, unreliable=False, reset=None):
        self.engine, self.group = engine, group
        self.n = engine.P
        self.net = Network(engine, group, reset)
        self.rpcs0 = engine.counters()["rpc_sent"]     # the counters are the engine's: a config reads what was added since it began
        self.bytes0 = {}
        self.applyErr = [""] * self.n
        self.rafts = [None] * self.n
        self.connected = [False] * self.n
        self.saved = [None] * self.n
        self.logs = [dict() for _ in range(self.n)]
        self.chans = [None] * self.n
        self.maxIndex = 0
        self.longreordering = False
        self.setunreliable(unreliable)
        for i in range(self.n):
            self.start1(i)
        for i in range(self.n):
            self.connect(i)
        self.t0 = engine.now()

    # ---- time ------------------------------------------------------------
    def advance(self, ticks):
        self.net.advance(ticks)
        self._reader()

    def _reader(self):
        """config.go:168-203"""
        for i in range(self.n):
            ch = self.chans[i]
            while ch is not None and not ch.empty():
                m = ch.get()
                if not m.CommandValid:
                    continue
                v = m.Command
                for j in range(self.n):
                    if m.CommandIndex in self.logs[j] and self.logs[j][m.CommandIndex] != v:
                        self.applyErr[i] = "commit index=%d server=%d %r != server=%d %r" % (
                            m.CommandIndex, i, v, j, self.logs[j][m.CommandIndex])
                prevok = (m.CommandIndex - 1) in self.logs[i]
                self.logs[i][m.CommandIndex] = v
                self.maxIndex = max(self.maxIndex, m.CommandIndex)
                if m.CommandIndex > 1 and not prevok:
                    self.applyErr[i] = "server %d apply out of order %d" % (i, m.CommandIndex)

    # ---- crash / restart (config.go:99-215) --------------------------------
    def crash1(self, i):
        self.disconnect(i)
        if self.rafts[i] is not None:
            self.engine.crash(self.group, i)      # cfg.net.DeleteServer(i), config.go:104: Calls no longer reach it
        if self.saved[i] is not None:
            self.saved[i] = self.saved[i].Copy()
        rf = self.rafts[i]
        if rf is not None:
            self.net.deliver()
            self._reader()
            rf.Kill()
            self.rafts[i] = None
        if self.saved[i] is not None:
            raftlog = self.saved[i].ReadRaftState()
            self.saved[i] = MakePersister()
            self.saved[i].SaveRaftState(raftlog)

    def start1(self, i):
        self.crash1(i)
        if self.saved[i] is not None:
            self.saved[i] = self.saved[i].Copy()
        else:
            self.saved[i] = MakePersister()
        self.chans[i] = queue.Queue()
        self.rafts[i] = Make(self.net.ends(i), i, self.saved[i], self.chans[i])

    # ---- connectivity (config.go:235-278) ----------------------------------
    def connect(self, i):
        self.connected[i] = self.net.connected[i] = True
        self.engine.set_connected(self.group, i, True)

    def disconnect(self, i):
        self.connected[i] = self.net.connected[i] = False
        self.engine.set_connected(self.group, i, False)

    # every config owns its network (config.go:70, 288-298): the rates are the group's, not the engine's
    def setunreliable(self, unrel):
        self.unreliable = unrel
        self._rates()

    def setlongreordering(self, yes):
        self.longreordering = yes
        self._rates()

    def _rates(self):
        drop = 100 if self.unreliable else 0
        self.engine.set_unreliable_group(self.group, drop, drop, 667 if self.longreordering else 0)

    def rpcTotal(self):
        return self.engine.counters()["rpc_sent"] - self.rpcs0

    def bytesTotal(self, command_bytes=8):
        """config.go:292-294; command_bytes = size of the commands the caller's handles stand for."""
        return self.engine.rpc_bytes(command_bytes) - self.bytes0.get(command_bytes, 0)

    def rpcCount(self):
        """Sum over servers of delivered requests (labrpc.go:406)."""
        return self.engine.counters()["rpc_delivered"]

    # ---- checks ------------------------------------------------------------
    def checkOneLeader(self):
        """config.go:302-332"""
        for _ in range(10):
            self.advance(50)
            leaders = {}
            for i in range(self.n):
                if self.connected[i]:
                    term, leader = self.rafts[i].GetState()
                    if leader:
                        leaders.setdefault(term, []).append(i)
            last = -1
            for term, ls in leaders.items():
                if len(ls) > 1:
                    raise TestFailure("term %d has %d (>1) leaders" % (term, len(ls)))
                last = max(last, term)
            if leaders:
                return leaders[last][0]
        raise TestFailure("expected one leader, got none")

    def checkTerms(self):
        """config.go:335-348"""
        term = -1
        for i in range(self.n):
            if self.connected[i]:
                xterm, _ = self.rafts[i].GetState()
                if term == -1:
                    term = xterm
                elif term != xterm:
                    raise TestFailure("servers disagree on term")
        return term

    def checkNoLeader(self):
        """config.go:351-360"""
        for i in range(self.n):
            if self.connected[i]:
                _, is_leader = self.rafts[i].GetState()
                if is_leader:
                    raise TestFailure("expected no leader, but %d claims to be leader" % i)

    def nCommitted(self, index):
        """config.go:363-385"""
        count, cmd = 0, None
        for i in range(self.n):
            if self.applyErr[i]:
                raise TestFailure(self.applyErr[i])
            if index in self.logs[i]:
                cmd1 = self.logs[i][index]
                if count > 0 and cmd != cmd1:
                    raise TestFailure("committed values do not match: index %d, %r, %r" % (index, cmd, cmd1))
                count += 1
                cmd = cmd1
        return count, cmd

    def wait(self, index, n, startTerm):
        """config.go:389-416"""
        to = 1
        for _ in range(30):
            nd, _ = self.nCommitted(index)
            if nd >= n:
                break
            self.advance(to)
            if to < 100:
                to *= 2
            if startTerm > -1:
                for r in self.rafts:
                    if r is not None:
                        t, _ = r.GetState()
                        if t > startTerm:
                            return -1
        nd, cmd = self.nCommitted(index)
        if nd < n:
            raise TestFailure("only %d decided for index %d; wanted %d" % (nd, index, n))
        return cmd

    def one(self, cmd, expectedServers, retry):
        """config.go:430-477"""
        t0 = self.engine.now()
        starts = 0
        while self.engine.now() - t0 < 1000:
            index = -1
            for _ in range(self.n):
                starts = (starts + 1) % self.n
                rf = self.rafts[starts] if self.connected[starts] else None
                if rf is not None:
                    index1, _, ok = rf.Start(cmd)
                    if ok:
                        index = index1
                        break
            if index != -1:
                t1 = self.engine.now()
                while self.engine.now() - t1 < 200:
                    nd, cmd1 = self.nCommitted(index)
                    if nd > 0 and nd >= expectedServers and cmd1 == cmd:
                        return index
                    self.advance(2)
                if not retry:
                    raise TestFailure("one(%r) failed to reach agreement" % (cmd,))
            else:
                self.advance(5)
        raise TestFailure("one(%r) failed to reach agreement" % (cmd,))
