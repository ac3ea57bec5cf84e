"""Scenario schedules: tests of src/raft/test_test.go written as per-tick op tables (rg_set_schedule) that EVERY
group of an engine executes on its own peers inside the tick kernel, so one launch runs the scenario on
thousands of groups at once.  Ticks are 10 ms; sleeps of the reference become gaps between ops."""
import random

from raft_amd._abi import (OP_CONNECT, OP_CRASH, OP_DISCONNECT, OP_PICK_LEADER, OP_PICK_OTHER, OP_PICK_RANDOM,
                           OP_RESTART_IF_FEWER, OP_START, OP_START_ALL, OP_START_ANY, SEL_ALL, SEL_REG)


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


ALL = {"Backup2B": backup_2b, "Figure82C": figure8_2c}
