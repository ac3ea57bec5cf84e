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
