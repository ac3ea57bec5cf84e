"""Scenario schedules: tests of src/raft/test_test.go written as per-tick op tables (rg_set_schedule) that EVERY
group of an engine executes on its own peers inside the tick kernel, so one launch runs the scenario on
thousands of groups at once.  Ticks are 10 ms; sleeps of the reference become gaps between ops."""
import random

from raft_amd._abi import (OP_CONNECT, OP_CONNECT_IF_FEWER, OP_CRASH, OP_DISCONNECT, OP_PICK_LEADER, OP_PICK_OTHER, OP_PICK_RANDOM,
                           OP_RESTART, OP_RESTART_IF_FEWER, OP_START, OP_START_ALL, OP_START_ANY, SEL_ALL, SEL_REG)


def backup_2b(gap=400):
    """TestBackup2B (test_test.go:371-441), 5 peers: the leader and one follower are cut off and pile up 50
    entries that cannot commit; the other three commit 50; then the new leader with one follower piles up 50 more;
    then the first pair plus that follower run alone and commit 50; finally everyone is reconnected.  `gap` ticks
    are left wherever the reference's cfg.one() would wait for a new leader."""
    ops, seq = [], [0]

    def cmd():
        seq[0] += 1
        return seq[0]
    t = 100
    ops.append((t, OP_START_ANY, 0, cmd()))                                    # cfg.one(rand, servers, true)
    t += 50
    ops.append((t, OP_PICK_LEADER, 0, 0))                                      # leader1 := checkOneLeader()
    ops += [(t, OP_DISCONNECT, SEL_REG(0, k), 0) for k in (2, 3, 4)]           # put leader and one follower in a partition
    ops += [(t + 1 + i, OP_START, SEL_REG(0, 0), cmd()) for i in range(50)]    # lots of commands that won't commit
    t += 60
    ops += [(t, OP_DISCONNECT, SEL_REG(0, k), 0) for k in (0, 1)]
    ops += [(t, OP_CONNECT, SEL_REG(0, k), 0) for k in (2, 3, 4)]              # allow other partition to recover
    t += gap
    ops += [(t + i, OP_START_ANY, 0, cmd()) for i in range(50)]                # lots of successful commands to new group
    t += 100
    ops.append((t, OP_PICK_LEADER, 1, 0))                                      # leader2
    ops.append((t, OP_PICK_OTHER, 2, SEL_REG(1, 0)))                           # other: a connected follower of leader2
    ops.append((t, OP_DISCONNECT, SEL_REG(2, 0), 0))
    ops += [(t + 1 + i, OP_START, SEL_REG(1, 0), cmd()) for i in range(50)]    # lots more commands that won't commit
    t += 60
    ops.append((t, OP_DISCONNECT, SEL_ALL, 0))                                 # bring original leader back to life
    ops += [(t, OP_CONNECT, SEL_REG(0, k), 0) for k in (0, 1)]
    ops.append((t, OP_CONNECT, SEL_REG(2, 0), 0))
    t += gap
    ops += [(t + i, OP_START_ANY, 0, cmd()) for i in range(50)]                # lots of successful commands
    t += 100
    ops.append((t, OP_CONNECT, SEL_ALL, 0))                                    # now everyone
    t += gap
    ops.append((t, OP_START_ANY, 0, cmd()))
    return ops, t + 100


def figure8_2c(iters=60, seed=8):
    """TestFigure82C (test_test.go:685-739), 5 peers: every iteration asks every live peer to Start a command,
    waits a short or a long while, crashes the peer that accepted, and restarts a random dead peer when fewer than
    three are alive; at the end all are restarted and one more command must go through."""
    rnd = random.Random(seed)
    ops, t = [(60, OP_START_ANY, 0, 1)], 120
    for i in range(iters):
        ops.append((t, OP_START_ALL, 0, 100 + i))
        t += rnd.randrange(0, 3) if rnd.randrange(1000) < 100 else rnd.randrange(0, 50)   # ms/10: RaftElectionTimeout/2 or 13 ms
        ops.append((t, OP_CRASH, SEL_REG(0, 0), 0))
        ops.append((t, OP_PICK_RANDOM, 1, i))
        ops.append((t, OP_RESTART_IF_FEWER, SEL_REG(1, 0), 3))
        t += 1
    ops.append((t, OP_RESTART_IF_FEWER, SEL_ALL, 6))                           # start1 + connect every dead server
    ops.append((t + 250, OP_START_ANY, 0, 9999))
    return ops, t + 400


def _one(ops, t, cmd, gap):
    """cfg.one(cmd, n, retry) (config.go:430-477): ask whoever leads, then leave `gap` ticks for the agreement."""
    ops.append((t, OP_START_ANY, 0, cmd))
    return t + gap


def reelection_2a(gap=300):
    """TestReElection2A (test_test.go:53-87), 3 peers.  Returns the tick of its checkNoLeader as well."""
    ops, t = [], 100
    ops += [(t, OP_PICK_LEADER, 0, 0), (t, OP_DISCONNECT, SEL_REG(0, 0), 0)]      # if the leader disconnects, a new one should be elected
    t += gap
    ops.append((t, OP_CONNECT, SEL_REG(0, 0), 0))                                 # the old leader rejoins
    t += 100
    ops += [(t, OP_PICK_LEADER, 1, 0), (t, OP_DISCONNECT, SEL_REG(1, 0), 0), (t, OP_DISCONNECT, SEL_REG(1, 1), 0)]   # no quorum
    t += 200                                                                      # 2 * RaftElectionTimeout
    no_leader_at = t
    ops.append((t, OP_CONNECT, SEL_REG(1, 1), 0))                                 # a quorum arises
    t += gap
    ops.append((t, OP_CONNECT, SEL_REG(1, 0), 0))                                 # re-join of the last node
    return ops, t + 100, no_leader_at


def fail_agree_2b(gap=60):
    """TestFailAgree2B (test_test.go:147-179), 3 peers: agreement despite a disconnected follower, which then catches up."""
    ops, t = [], 100
    t = _one(ops, t, 101, gap)
    ops += [(t, OP_PICK_LEADER, 0, 0), (t, OP_DISCONNECT, SEL_REG(0, 1), 0)]
    t = _one(ops, t + 1, 102, gap)
    t = _one(ops, t, 103, gap)
    t += 100                                                                      # RaftElectionTimeout
    t = _one(ops, t, 104, gap)
    t = _one(ops, t, 105, gap)
    ops.append((t, OP_CONNECT, SEL_REG(0, 1), 0))
    t += 300                                                                      # the returning peer's term forces an election; one() retries
    t = _one(ops, t, 106, gap)
    t += 100
    t = _one(ops, t, 107, 2 * gap)
    return ops, t, 7


def fail_no_agree_2b(gap=300):
    """TestFailNoAgree2B (test_test.go:181-230), 5 peers: three followers leave, the leader's entry must not commit;
    after the repair the group agrees again."""
    ops, t = [], 100
    t = _one(ops, t, 10, 60)
    ops.append((t, OP_PICK_LEADER, 0, 0))
    ops += [(t, OP_DISCONNECT, SEL_REG(0, k), 0) for k in (1, 2, 3)]
    ops.append((t + 1, OP_START, SEL_REG(0, 0), 20))
    t += 200                                                                      # 2 * RaftElectionTimeout
    ops += [(t, OP_CONNECT, SEL_REG(0, k), 0) for k in (1, 2, 3)]
    t += gap                                                                      # checkOneLeader
    t = _one(ops, t, 30, 60)
    t = _one(ops, t, 1000, 100)
    return ops, t, None


def rejoin_2b(gap=300):
    """TestRejoin2B (test_test.go:333-369), 3 peers: a partitioned leader piles up entries, its successor is
    partitioned in turn, the old leader comes back."""
    ops, t = [], 100
    t = _one(ops, t, 101, 60)
    ops += [(t, OP_PICK_LEADER, 0, 0), (t, OP_DISCONNECT, SEL_REG(0, 0), 0)]
    ops += [(t + 1 + i, OP_START, SEL_REG(0, 0), 102 + i) for i in range(3)]      # the old leader tries to agree on some entries
    t += gap
    t = _one(ops, t, 103, 60)                                                     # new leader commits, also for index 2
    ops += [(t, OP_PICK_LEADER, 1, 0), (t, OP_DISCONNECT, SEL_REG(1, 0), 0), (t, OP_CONNECT, SEL_REG(0, 0), 0)]
    t += gap
    t = _one(ops, t, 104, 60)
    ops.append((t, OP_CONNECT, SEL_REG(1, 0), 0))                                 # all together now
    t += gap
    t = _one(ops, t, 105, 100)
    return ops, t, 4


def persist1_2c(gap=300):
    """TestPersist12C (test_test.go:553-597), 3 peers: everyone is crashed and restarted, then leaders and a follower
    in turn, with agreements in between."""
    ops, t = [], 100
    t = _one(ops, t, 11, 60)
    ops += [(t, OP_RESTART, k, 0) for k in range(3)] + [(t, OP_CONNECT, k, 0) for k in range(3)]   # crash and re-start all
    t += gap
    t = _one(ops, t, 12, 60)
    ops += [(t, OP_PICK_LEADER, 0, 0), (t, OP_RESTART, SEL_REG(0, 0), 0), (t, OP_CONNECT, SEL_REG(0, 0), 0)]
    t += gap
    t = _one(ops, t, 13, 60)
    ops += [(t, OP_PICK_LEADER, 1, 0), (t, OP_DISCONNECT, SEL_REG(1, 0), 0)]
    t += gap
    t = _one(ops, t, 14, 60)
    ops += [(t, OP_RESTART, SEL_REG(1, 0), 0), (t, OP_CONNECT, SEL_REG(1, 0), 0)]
    t += gap                                                                      # cfg.wait(4, servers, -1)
    ops += [(t, OP_PICK_LEADER, 2, 0), (t, OP_DISCONNECT, SEL_REG(2, 1), 0)]      # i3 = leader + 1
    t = _one(ops, t + 1, 15, 60)
    ops += [(t, OP_RESTART, SEL_REG(2, 1), 0), (t, OP_CONNECT, SEL_REG(2, 1), 0)]
    t += 60
    t = _one(ops, t, 16, 100)
    return ops, t, 6


def persist2_2c(gap=300, iters=5):
    """TestPersist22C (test_test.go:599-643), 5 peers: rolling disconnects and restarts around three agreements per round."""
    ops, t, index = [], 100, 1
    for _ in range(iters):
        t = _one(ops, t, 10 + index, 60)
        index += 1
        ops.append((t, OP_PICK_LEADER, 0, 0))
        ops += [(t, OP_DISCONNECT, SEL_REG(0, k), 0) for k in (1, 2)]
        t = _one(ops, t + 1, 10 + index, 60)
        index += 1
        ops += [(t, OP_DISCONNECT, SEL_REG(0, k), 0) for k in (0, 3, 4)]
        ops += [(t, OP_RESTART, SEL_REG(0, k), 0) for k in (1, 2)] + [(t, OP_CONNECT, SEL_REG(0, k), 0) for k in (1, 2)]
        t += 100                                                                  # RaftElectionTimeout
        ops += [(t, OP_RESTART, SEL_REG(0, 3), 0), (t, OP_CONNECT, SEL_REG(0, 3), 0)]
        t += gap
        t = _one(ops, t, 10 + index, 60)
        index += 1
        ops += [(t, OP_CONNECT, SEL_REG(0, 4), 0), (t, OP_CONNECT, SEL_REG(0, 0), 0)]
        t += gap
    t = _one(ops, t, 1000, 100)
    return ops, t, index


def persist3_2c(gap=300):
    """TestPersist32C (test_test.go:645-673), 3 peers: partitioned leader and one follower crash, the leader restarts."""
    ops, t = [], 100
    t = _one(ops, t, 101, 60)
    ops += [(t, OP_PICK_LEADER, 0, 0), (t, OP_DISCONNECT, SEL_REG(0, 2), 0)]
    t = _one(ops, t + 1, 102, 60)
    ops += [(t, OP_CRASH, SEL_REG(0, 0), 0), (t, OP_CRASH, SEL_REG(0, 1), 0), (t, OP_CONNECT, SEL_REG(0, 2), 0),
            (t, OP_RESTART, SEL_REG(0, 0), 0), (t, OP_CONNECT, SEL_REG(0, 0), 0)]
    t += gap
    t = _one(ops, t, 103, 60)
    ops += [(t, OP_RESTART, SEL_REG(0, 1), 0), (t, OP_CONNECT, SEL_REG(0, 1), 0)]
    t += 100
    t = _one(ops, t, 104, 100)
    return ops, t, 4


def figure8_unreliable_2c(iters=400, seed=9):
    """TestFigure8Unreliable2C (test_test.go:770-823), 5 peers on an unreliable net: every iteration asks every peer
    to Start a command, waits a short or (one time in ten) a long while, disconnects the connected leader half of the
    time, and reconnects a random peer when fewer than three are connected; at the end all are reconnected and one
    more command must go through.  Returns the tick of the iteration at which the reference turns long reordering on
    (a host verb: cfg.setlongreordering, config.go:296-298) as well."""
    rnd = random.Random(seed)
    ops, t, reorder_at = [(60, OP_START_ANY, 0, 1)], 120, None
    for i in range(iters):
        if i == iters // 5:
            reorder_at = t
        ops.append((t, OP_START_ALL, 3, 100 + i))
        ops.append((t, OP_PICK_LEADER, 0, 0))                                   # a connected peer that accepted
        t += rnd.randrange(0, 50) if rnd.randrange(1000) < 100 else rnd.randrange(0, 13) // 10 + (rnd.randrange(10) < 6)
        if rnd.randrange(1000) < 500:
            ops.append((t, OP_DISCONNECT, SEL_REG(0, 0), 0))
        ops.append((t, OP_PICK_RANDOM, 1, i))
        ops.append((t, OP_CONNECT_IF_FEWER, SEL_REG(1, 0), 3))
    ops.append((t, OP_CONNECT, SEL_ALL, 0))
    ops.append((t + 400, OP_START_ANY, 0, 9999))
    return ops, t + 600, reorder_at


def unreliable_agree_2c(iters=49):
    """TestUnreliableAgree2C (test_test.go:741-768), 5 peers on an unreliable net: five concurrent cfg.one() per
    iteration; the reference then makes the net reliable (a host verb) and wants one more agreement on all servers.
    Returns the tick at which the net becomes reliable."""
    ops, t = [], 150
    for it in range(1, iters + 1):
        ops += [(t, OP_START_ANY, 0, 100 * it + j) for j in range(4)] + [(t, OP_START_ANY, 0, it)]
        t += 8
    reliable_at = t
    ops.append((t + 300, OP_START_ANY, 0, 100))
    return ops, t + 500, reliable_at


def basic_agree_2b():
    """TestBasicAgree2B (test_test.go:89-114), 3 peers: three agreements get indices 1, 2, 3 on every server."""
    ops, t = [], 100
    for i in range(3):
        t = _one(ops, t, 100 * (i + 1), 40)
    return ops, t, 3


def concurrent_starts_2b():
    """TestConcurrentStarts2B (test_test.go:232-331), 3 peers: Start(1) and five more Starts on the leader at once;
    all six commit on every server in the leader's term."""
    ops, t = [], 100
    ops.append((t, OP_PICK_LEADER, 0, 0))
    ops.append((t, OP_START, SEL_REG(0, 0), 1))
    ops += [(t, OP_START, SEL_REG(0, 0), 100 + i) for i in range(5)]
    return ops, t + 100, 6


# name -> (peers, schedule); the schedules return (ops, ticks, commitIndex every peer should end on, or None)
MORE = {"BasicAgree2B": (3, basic_agree_2b), "ConcurrentStarts2B": (3, concurrent_starts_2b), "FailAgree2B": (3, fail_agree_2b), "FailNoAgree2B": (5, fail_no_agree_2b), "Rejoin2B": (3, rejoin_2b),
        "Persist12C": (3, persist1_2c), "Persist22C": (5, persist2_2c), "Persist32C": (3, persist3_2c)}

ALL = {"Backup2B": backup_2b, "Figure82C": figure8_2c}
