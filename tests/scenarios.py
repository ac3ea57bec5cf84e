"""The reference's Lab 2A/2B/2C tests in tick units (reference:
src/raft/test_test.go).  Each scenario takes `make_engine(**config)` so the
same script runs against the CPU checker and, under -m gpu, the HIP engine.
Sleeps become advance(ticks) at 10 ms per tick; rand becomes a seeded
random.Random."""
import random

from raft_amd.harness import Config, TestFailure, ELECTION_TIMEOUT_TICKS as ET


def _cfg(make_engine, servers, unreliable=False, log_cap=512, seed=1):
    eng = make_engine(n_groups=1, n_peers=servers, log_cap=log_cap, seed=seed, account_bytes=1, apply_ring_cap=log_cap)
    return Config(eng, 0, unreliable)


def initial_election_2a(make_engine):          # test_test.go:22-51
    cfg = _cfg(make_engine, 3)
    cfg.checkOneLeader()
    cfg.advance(5)
    term1 = cfg.checkTerms()
    assert term1 >= 1
    cfg.advance(2 * ET)
    term2 = cfg.checkTerms()
    assert term1 == term2        # the reference only warns; lockstep without failures must hold it
    cfg.checkOneLeader()


def reelection_2a(make_engine):                # test_test.go:53-87
    servers = 3
    cfg = _cfg(make_engine, servers)
    leader1 = cfg.checkOneLeader()
    cfg.disconnect(leader1)
    cfg.checkOneLeader()
    cfg.connect(leader1)
    leader2 = cfg.checkOneLeader()
    cfg.disconnect(leader2)
    cfg.disconnect((leader2 + 1) % servers)
    cfg.advance(2 * ET)
    cfg.checkNoLeader()
    cfg.connect((leader2 + 1) % servers)
    cfg.checkOneLeader()
    cfg.connect(leader2)
    cfg.checkOneLeader()


def basic_agree_2b(make_engine):               # test_test.go:89-110
    servers = 3
    cfg = _cfg(make_engine, servers)
    for index in range(1, 4):
        nd, _ = cfg.nCommitted(index)
        assert nd == 0, "some have committed before Start()"
        xindex = cfg.one(index * 100, servers, False)
        assert xindex == index, "got index %d but expected %d" % (xindex, index)
    return cfg


def rpc_bytes_2b(make_engine):                 # test_test.go:112-145
    """Each command travels to each peer just once: ten 5000-byte commands (the engine carries handles, the byte
    count prices each entry at the size the caller states) cost at most servers x 50,000 bytes + 50,000."""
    servers, size = 3, 5000
    cfg = _cfg(make_engine, servers)
    cfg.one(99, servers, False)
    bytes0 = cfg.bytesTotal(size)
    sent = 0
    for index in range(2, 12):
        xindex = cfg.one(1000 + index, servers, False)
        assert xindex == index, "got index %d but expected %d" % (xindex, index)
        sent += size
    got = cfg.bytesTotal(size) - bytes0
    expected = servers * sent
    assert got <= expected + 50000, "too many RPC bytes; got %d, expected %d" % (got, expected)
    assert got >= (servers - 1) * sent, "fewer bytes than one copy per follower (%d)" % got
    return cfg


def fail_agree_2b(make_engine):                # test_test.go:147-179
    servers = 3
    cfg = _cfg(make_engine, servers)
    cfg.one(101, servers, False)
    leader = cfg.checkOneLeader()
    cfg.disconnect((leader + 1) % servers)
    cfg.one(102, servers - 1, False)
    cfg.one(103, servers - 1, False)
    cfg.advance(ET)
    cfg.one(104, servers - 1, False)
    cfg.one(105, servers - 1, False)
    cfg.connect((leader + 1) % servers)
    cfg.one(106, servers, True)
    cfg.advance(ET)
    cfg.one(107, servers, True)


def fail_no_agree_2b(make_engine):             # test_test.go:181-230
    servers = 5
    cfg = _cfg(make_engine, servers)
    cfg.one(10, servers, False)
    leader = cfg.checkOneLeader()
    for d in (1, 2, 3):
        cfg.disconnect((leader + d) % servers)
    index, _, ok = cfg.rafts[leader].Start(20)
    assert ok, "leader rejected Start()"
    assert index == 2, "expected index 2, got %d" % index
    cfg.advance(2 * ET)
    n, _ = cfg.nCommitted(index)
    assert n == 0, "%d committed but no majority" % n
    for d in (1, 2, 3):
        cfg.connect((leader + d) % servers)
    leader2 = cfg.checkOneLeader()
    index2, _, ok2 = cfg.rafts[leader2].Start(30)
    assert ok2, "leader2 rejected Start()"
    assert 2 <= index2 <= 3, "unexpected index %d" % index2
    cfg.one(1000, servers, True)


def concurrent_starts_2b(make_engine):         # test_test.go:232-331 (Starts issued within one tick)
    servers = 3
    cfg = _cfg(make_engine, servers)
    for attempt in range(5):
        if attempt > 0:
            cfg.advance(300)
        leader = cfg.checkOneLeader()
        _, term, ok = cfg.rafts[leader].Start(1)
        if not ok:
            continue
        idxs = []
        for ii in range(5):
            i, term1, ok = cfg.rafts[leader].Start(100 + ii)
            if term1 == term and ok:
                idxs.append(i)
        if any(cfg.rafts[j].GetState()[0] != term for j in range(servers)):
            continue
        cmds, failed = [], False
        for index in idxs:
            cmd = cfg.wait(index, servers, term)
            if cmd == -1:
                failed = True
                break
            cmds.append(cmd)
        if failed:
            continue
        for ii in range(5):
            assert 100 + ii in cmds, "cmd %d missing in %r" % (100 + ii, cmds)
        return
    raise TestFailure("term changed too often")


def rejoin_2b(make_engine):                    # test_test.go:333-369
    servers = 3
    cfg = _cfg(make_engine, servers)
    cfg.one(101, servers, True)
    leader1 = cfg.checkOneLeader()
    cfg.disconnect(leader1)
    cfg.rafts[leader1].Start(102)
    cfg.rafts[leader1].Start(103)
    cfg.rafts[leader1].Start(104)
    cfg.one(103, 2, True)
    leader2 = cfg.checkOneLeader()
    cfg.disconnect(leader2)
    cfg.connect(leader1)
    cfg.one(104, 2, True)
    cfg.connect(leader2)
    cfg.one(105, servers, True)


def backup_2b(make_engine, n=50):              # test_test.go:371-441
    servers = 5
    rnd = random.Random(5)
    cfg = _cfg(make_engine, servers)
    cfg.one(rnd.getrandbits(60), servers, True)
    leader1 = cfg.checkOneLeader()
    for d in (2, 3, 4):
        cfg.disconnect((leader1 + d) % servers)
    for _ in range(n):
        cfg.rafts[leader1].Start(rnd.getrandbits(60))
    cfg.advance(ET // 2)
    cfg.disconnect((leader1 + 0) % servers)
    cfg.disconnect((leader1 + 1) % servers)
    for d in (2, 3, 4):
        cfg.connect((leader1 + d) % servers)
    for _ in range(n):
        cfg.one(rnd.getrandbits(60), 3, True)
    leader2 = cfg.checkOneLeader()
    other = (leader1 + 2) % servers
    if leader2 == other:
        other = (leader2 + 1) % servers
    cfg.disconnect(other)
    for _ in range(n):
        cfg.rafts[leader2].Start(rnd.getrandbits(60))
    cfg.advance(ET // 2)
    for i in range(servers):
        cfg.disconnect(i)
    cfg.connect((leader1 + 0) % servers)
    cfg.connect((leader1 + 1) % servers)
    cfg.connect(other)
    for _ in range(n):
        cfg.one(rnd.getrandbits(60), 3, True)
    for i in range(servers):
        cfg.connect(i)
    cfg.one(rnd.getrandbits(60), servers, True)
    return cfg


def count_2b(make_engine):                     # test_test.go:443-551
    servers = 3
    rnd = random.Random(7)
    cfg = _cfg(make_engine, servers)
    leader = cfg.checkOneLeader()
    total1 = cfg.rpcCount()
    assert 1 <= total1 <= 30, "too many or few RPCs (%d) to elect initial leader" % total1
    total2 = None
    for attempt in range(5):
        if attempt > 0:
            cfg.advance(300)
        leader = cfg.checkOneLeader()
        total1 = cfg.rpcCount()
        iters = 10
        starti, term, ok = cfg.rafts[leader].Start(1)
        if not ok:
            continue
        cmds, again = [], False
        for i in range(1, iters + 2):
            x = rnd.getrandbits(31)
            cmds.append(x)
            index1, term1, ok = cfg.rafts[leader].Start(x)
            if term1 != term or not ok:
                again = True
                break
            assert starti + i == index1, "Start() failed"
        if again:
            continue
        for i in range(1, iters + 1):
            cmd = cfg.wait(starti + i, servers, term)
            if cmd == -1:
                again = True
                break
            assert cmd == cmds[i - 1], "wrong value %r committed for index %d; expected %r" % (cmd, starti + i, cmds)
        if again:
            continue
        failed = any(cfg.rafts[j].GetState()[0] != term for j in range(servers))
        total2 = cfg.rpcCount()
        if failed:
            continue
        assert total2 - total1 <= (iters + 1 + 3) * 3, "too many RPCs (%d) for %d entries" % (total2 - total1, iters)
        break
    else:
        raise TestFailure("term changed too often")
    cfg.advance(ET)
    total3 = cfg.rpcCount()
    assert total3 - total2 <= 3 * 20, "too many RPCs (%d) for 1 second of idleness" % (total3 - total2)


def persist1_2c(make_engine):                  # test_test.go:553-597
    servers = 3
    cfg = _cfg(make_engine, servers)
    cfg.one(11, servers, True)
    for i in range(servers):
        cfg.start1(i)
    for i in range(servers):
        cfg.disconnect(i)
        cfg.connect(i)
    cfg.one(12, servers, True)
    leader1 = cfg.checkOneLeader()
    cfg.disconnect(leader1)
    cfg.start1(leader1)
    cfg.connect(leader1)
    cfg.one(13, servers, True)
    leader2 = cfg.checkOneLeader()
    cfg.disconnect(leader2)
    cfg.one(14, servers - 1, True)
    cfg.start1(leader2)
    cfg.connect(leader2)
    cfg.wait(4, servers, -1)
    i3 = (cfg.checkOneLeader() + 1) % servers
    cfg.disconnect(i3)
    cfg.one(15, servers - 1, True)
    cfg.start1(i3)
    cfg.connect(i3)
    cfg.one(16, servers, True)


def persist2_2c(make_engine, iters=5):         # test_test.go:599-643
    servers = 5
    cfg = _cfg(make_engine, servers)
    index = 1
    for _ in range(iters):
        cfg.one(10 + index, servers, True)
        index += 1
        leader1 = cfg.checkOneLeader()
        cfg.disconnect((leader1 + 1) % servers)
        cfg.disconnect((leader1 + 2) % servers)
        cfg.one(10 + index, servers - 2, True)
        index += 1
        for d in (0, 3, 4):
            cfg.disconnect((leader1 + d) % servers)
        cfg.start1((leader1 + 1) % servers)
        cfg.start1((leader1 + 2) % servers)
        cfg.connect((leader1 + 1) % servers)
        cfg.connect((leader1 + 2) % servers)
        cfg.advance(ET)
        cfg.start1((leader1 + 3) % servers)
        cfg.connect((leader1 + 3) % servers)
        cfg.one(10 + index, servers - 2, True)
        index += 1
        cfg.connect((leader1 + 4) % servers)
        cfg.connect((leader1 + 0) % servers)
    cfg.one(1000, servers, True)


def persist3_2c(make_engine):                  # test_test.go:645-673
    servers = 3
    cfg = _cfg(make_engine, servers)
    cfg.one(101, 3, True)
    leader = cfg.checkOneLeader()
    cfg.disconnect((leader + 2) % servers)
    cfg.one(102, 2, True)
    cfg.crash1((leader + 0) % servers)
    cfg.crash1((leader + 1) % servers)
    cfg.connect((leader + 2) % servers)
    cfg.start1((leader + 0) % servers)
    cfg.connect((leader + 0) % servers)
    cfg.one(103, 2, True)
    cfg.start1((leader + 1) % servers)
    cfg.connect((leader + 1) % servers)
    cfg.one(104, servers, True)


def figure8_2c(make_engine, iters=200, seed=3):   # test_test.go:685-739
    servers = 5
    rnd = random.Random(seed)
    cfg = _cfg(make_engine, servers, log_cap=4096)
    cfg.one(rnd.getrandbits(60), 1, True)
    nup = servers
    for _ in range(iters):
        leader = -1
        for i in range(servers):
            if cfg.rafts[i] is not None:
                _, _, ok = cfg.rafts[i].Start(rnd.getrandbits(60))
                if ok:
                    leader = i
        if rnd.randrange(1000) < 100:
            cfg.advance(rnd.randrange(ET // 2))
        else:
            cfg.advance(rnd.randrange(2))          # 0..12 ms
        if leader != -1:
            cfg.crash1(leader)
            nup -= 1
        if nup < 3:
            s = rnd.randrange(servers)
            if cfg.rafts[s] is None:
                cfg.start1(s)
                cfg.connect(s)
                nup += 1
    for i in range(servers):
        if cfg.rafts[i] is None:
            cfg.start1(i)
            cfg.connect(i)
    cfg.one(rnd.getrandbits(60), servers, True)


def unreliable_agree_2c(make_engine, iters=50):   # test_test.go:741-768 (the 5 concurrent one()s run in turn)
    servers = 5
    cfg = _cfg(make_engine, servers, unreliable=True, log_cap=1024)
    for it in range(1, iters):
        for j in range(4):
            cfg.one(100 * it + j, 1, True)
        cfg.one(it, 1, True)
    cfg.setunreliable(False)
    cfg.one(100, servers, True)


def figure8_unreliable_2c(make_engine, iters=300, seed=4):   # test_test.go:770-823
    servers = 5
    rnd = random.Random(seed)
    cfg = _cfg(make_engine, servers, unreliable=True, log_cap=4096)
    cfg.one(rnd.randrange(10000), 1, True)
    nup = servers
    for it in range(iters):
        if it == iters // 5:
            cfg.setlongreordering(True)
        leader = -1
        for i in range(servers):
            _, _, ok = cfg.rafts[i].Start(rnd.randrange(10000))
            if ok and cfg.connected[i]:
                leader = i
        if rnd.randrange(1000) < 100:
            cfg.advance(rnd.randrange(ET // 2))
        else:
            cfg.advance(rnd.randrange(2))
        if leader != -1 and rnd.randrange(1000) < ET * 10 // 2:
            cfg.disconnect(leader)
            nup -= 1
        if nup < 3:
            s = rnd.randrange(servers)
            if not cfg.connected[s]:
                cfg.connect(s)
                nup += 1
    for i in range(servers):
        if not cfg.connected[i]:
            cfg.connect(i)
    cfg.setlongreordering(False)
    cfg.one(rnd.randrange(10000), servers, True)


def churn_2c(make_engine, unreliable, seed=6, iters=20):   # test_test.go:825-968 (clients interleaved with the churn loop)
    servers = 5
    rnd = random.Random(seed)
    cfg = _cfg(make_engine, servers, unreliable=unreliable, log_cap=4096)
    values = []

    def client_round():
        x = rnd.getrandbits(60)
        index, ok = -1, False
        for i in range(servers):
            rf = cfg.rafts[i]
            if rf is not None:
                index1, _, ok1 = rf.Start(x)
                if ok1:
                    ok, index = True, index1
        if ok:
            for to in (1, 2, 5, 10, 20):
                nd, cmd = cfg.nCommitted(index)
                if nd > 0:
                    if cmd == x:
                        values.append(x)
                    break
                cfg.advance(to)
        else:
            cfg.advance(8)

    for _ in range(iters):
        if rnd.randrange(1000) < 200:
            cfg.disconnect(rnd.randrange(servers))
        if rnd.randrange(1000) < 500:
            i = rnd.randrange(servers)
            if cfg.rafts[i] is None:
                cfg.start1(i)
            cfg.connect(i)
        if rnd.randrange(1000) < 200:
            i = rnd.randrange(servers)
            if cfg.rafts[i] is not None:
                cfg.crash1(i)
        t0 = cfg.engine.now()
        while cfg.engine.now() - t0 < ET * 7 // 10:
            client_round()
    cfg.advance(ET)
    cfg.setunreliable(False)
    for i in range(servers):
        if cfg.rafts[i] is None:
            cfg.start1(i)
        cfg.connect(i)
    cfg.advance(ET)
    lastIndex = cfg.one(rnd.getrandbits(60), servers, True)
    really = [cfg.wait(index, servers, -1) for index in range(1, lastIndex + 1)]
    for v1 in values:
        assert v1 in really, "didn't find a value"


ALL = {
    "InitialElection2A": initial_election_2a, "ReElection2A": reelection_2a, "BasicAgree2B": basic_agree_2b,
    "RPCBytes2B": rpc_bytes_2b, "FailAgree2B": fail_agree_2b, "FailNoAgree2B": fail_no_agree_2b, "ConcurrentStarts2B": concurrent_starts_2b,
    "Rejoin2B": rejoin_2b, "Backup2B": backup_2b, "Count2B": count_2b, "Persist12C": persist1_2c,
    "Persist22C": persist2_2c, "Persist32C": persist3_2c, "Figure82C": figure8_2c,
    "UnreliableAgree2C": unreliable_agree_2c, "Figure8Unreliable2C": figure8_unreliable_2c,
    "ReliableChurn2C": lambda mk: churn_2c(mk, False), "UnreliableChurn2C": lambda mk: churn_2c(mk, True),
}


# ---- several tester configs on one engine, and the labrpc-shaped ends -------------------------------------------

def two_configs_on_one_engine(make_engine):
    """config.go:60-96: every make_config builds fresh peers on a network of its own.  Two configs share one
    engine here: the second is made 3 s after the first, its peers are fresh Makes at that moment (term 0, peer 0
    with timeout 0 wins term 1, raft.go:545), and its network is reliable while the first one's is not."""
    eng = make_engine(n_groups=4, n_peers=3, log_cap=128, seed=3, account_bytes=1, apply_ring_cap=128)
    a = Config(eng, 0, unreliable=True)
    for i in range(3):
        a.one(100 + i, 3, True)
    a.advance(300 - (eng.now() % 300))
    t_b = eng.now()
    assert eng.read("term")[2].max() >= 1                   # group 2 has been running all along ...
    b = Config(eng, 2, unreliable=False)
    assert eng.read("term")[2].tolist() == [0, 0, 0]        # ... and starts over for its new config
    assert eng.read("loglen")[2].tolist() == [1, 1, 1] and eng.read("timer")[2].tolist() == [t_b * 10] * 3
    assert eng.read("term")[0].min() >= 1                   # the first config is not touched
    assert b.checkOneLeader() == 0 and b.checkTerms() == 1
    before = eng.counters()
    assert b.one(11, 3, False) == 1
    assert a.one(200, 3, True) == 4
    assert b.one(12, 3, False) == 2
    assert b.logs[0] == {1: 11, 2: 12} and [a.logs[0][i] for i in (1, 2, 3, 4)] == [100, 101, 102, 200]
    assert eng.safety_check() == (0, 0)
    return eng


def labrpc_ends_reach_the_handlers(make_engine):
    """peers[i].Call("Raft.RequestVote" | "Raft.AppendEntries", args, reply) (labrpc.go:84-121) on the ends of one
    config: the handler of the target peer runs (raft.go:200-232, 270-321), a disconnected end loses the request,
    and the reply is decoded like gob does: zero fields do not overwrite (labgob.go:116-121)."""
    from raft_amd.raft import AppendEntriesArgs, AppendEntriesReply, RequestVoteArgs, RequestVoteReply
    eng = make_engine(n_groups=2, n_peers=3, log_cap=64, seed=5, apply_ring_cap=64)
    cfg = Config(eng, 1, False)
    leader = cfg.checkOneLeader()
    assert leader == 0
    ends = cfg.net.ends(2)                                  # server 2's ends
    assert len(ends) == 3
    # a vote request of a higher term: granted, and the leader it reaches is demoted (raft.go:208-216)
    reply = RequestVoteReply(Term=0, VoteGranted=False)
    assert ends[0].Call("Raft.RequestVote", RequestVoteArgs(5, 2, 0, 0), reply) is True
    assert (reply.Term, reply.VoteGranted) == (5, True)
    assert cfg.rafts[0].GetState() == (5, False)
    # the same struct reused (raft.go:585-588): a refusal leaves VoteGranted as it was -- false is not transmitted
    assert ends[1].Call("Raft.RequestVote", RequestVoteArgs(5, 1, 0, 0), reply) is True     # peer 1 has voted for 0 in term 1, no vote yet in 5
    reply2 = RequestVoteReply(Term=5, VoteGranted=True)
    assert ends[1].Call("Raft.RequestVote", RequestVoteArgs(5, 2, 0, 0), reply2) is True    # voted for 1 just now: refuses 2
    assert reply2.VoteGranted is True
    # a disconnected end: the request is lost, nothing is decoded
    cfg.disconnect(1)
    lost = RequestVoteReply(Term=-7, VoteGranted=False)
    assert ends[1].Call("Raft.RequestVote", RequestVoteArgs(9, 2, 0, 0), lost) is False
    assert (lost.Term, lost.VoteGranted) == (-7, False)
    assert cfg.rafts[1].GetState()[0] == 5
    cfg.connect(1)
    # AppendEntries with arbitrary commands: appended, committed, applied as the objects they were
    ar = AppendEntriesReply()
    ok = ends[0].Call("Raft.AppendEntries", AppendEntriesArgs(6, 2, 0, 0, [(6, "x"), (6, ("y", 2))], 2), ar)
    assert ok and (ar.Term, ar.Success) == (6, True)
    assert eng.read("loglen")[1, 0] == 3 and eng.read("commit")[1, 0] == 2 and eng.read("vote")[1, 0] == 2
    cfg.advance(2)
    assert cfg.logs[0] == {1: "x", 2: ("y", 2)}
    # a conflict hint (raft.go:303-314): PrevLogIndex past the end
    ar = AppendEntriesReply()
    assert ends[1].Call("Raft.AppendEntries", AppendEntriesArgs(6, 2, 7, 6, [], 0), ar)
    assert (ar.Success, ar.ConflictTerm, ar.ConflictFirstIndex) == (False, -1, 1)
    assert ends[1].Call("Raft.Nope", None, None) is False
    return eng


def parked_slots_stay_silent(make_engine):
    """The Go binding keeps one 4,096-group engine per peer count and gives every tester config a group of it; its
    rpcTotal / bytesTotal are deltas of the ENGINE's counters (config.go:284-294).  Groups no config has taken are
    parked (every peer crashed and off the net), so the counts a config reads are its own: TestCount2B's ceilings
    (test_test.go:461, 527, 546) hold on a 4,096-group engine as they do on a one-group engine."""
    from raft_amd.harness import park_all
    G = 4096
    eng = make_engine(n_groups=G, n_peers=3, log_cap=64, seed=7, account_bytes=1, apply_ring_cap=64)
    park_all(eng)
    eng.step(150)
    c = eng.counters()
    assert (c["rpc_sent"], c["rv_rounds"], c["timeouts"], c["elections_won"]) == (0, 0, 0, 0)
    cfg = Config(eng, 77, False, reset=True)
    cfg.checkOneLeader()
    total1 = cfg.rpcTotal()
    assert 1 <= total1 <= 30, "too many or few RPCs (%d) to elect initial leader" % total1        # test_test.go:461
    for i in range(10):
        assert cfg.one(500 + i, 3, False) == i + 1
    total2 = cfg.rpcTotal()
    assert total2 - total1 <= (10 + 1 + 3) * 3, "too many RPCs (%d) for 10 entries" % (total2 - total1)   # test_test.go:527
    cfg.advance(ET)
    total3 = cfg.rpcTotal()
    assert total3 - total2 <= 3 * 20, "too many RPCs (%d) for 1 second of idleness" % (total3 - total2)   # test_test.go:546
    assert 0 < cfg.bytesTotal(8) < 4000
    rounds = eng.read("roundseq").sum(axis=1)
    assert rounds[77] > 0 and rounds.sum() == rounds[77]      # nobody else sent anything
    assert eng.read("term")[:77].max() == 0 and eng.read("term")[78:].max() == 0
    return eng


EXTRA = {"TwoConfigsOneEngine": two_configs_on_one_engine, "LabrpcEnds": labrpc_ends_reach_the_handlers,
         "ParkedSlots": parked_slots_stay_silent}
