"""Hand-derived known answers, each a function of an engine factory so that the
CPU checker and the HIP engine are held to the same vectors.  Derivations cite
the reference statements they follow (src/raft/raft.go)."""
import numpy as np

from raft_amd import _abi


def _set_log(e, g, p, terms, cmds=None, commit=None, term=None):
    L = e.L
    lt, lc, ln = e.read("logterm"), e.read("logcmd"), e.read("loglen")
    lt[g, p, :len(terms)] = terms
    lc[g, p, :len(terms)] = cmds if cmds is not None else [100 * i for i in range(len(terms))]
    ln[g, p] = len(terms)
    e.write("logterm", lt)
    e.write("logcmd", lc)
    e.write("loglen", ln)
    if commit is not None:
        c = e.read("commit")
        c[g, p] = commit
        e.write("commit", c)
    if term is not None:
        t = e.read("term")
        t[g, p] = term
        e.write("term", t)


def boot_and_three_starts(mk):
    """Lossless 3-peer boot: peer 0 starts with timeout 0 (raft.go:545) and wins
    term 1; three Starts get indices 1,2,3 (test_test.go:97-106) and every peer
    ends with terms [0,1,1,1], commitIndex 3, voteFor 0."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.step(30)
    assert e.get_state(0, 0) == (1, True)
    assert [e.get_state(0, p)[1] for p in (1, 2)] == [False, False]
    for i, cmd in enumerate((100, 200, 300)):
        assert e.start(0, 0, cmd) == (i + 1, 1, True)
        e.step(12)
    assert e.read("logterm")[0, :, :4].tolist() == [[0, 1, 1, 1]] * 3
    assert e.read("logcmd")[0, :, 1:4].tolist() == [[100, 200, 300]] * 3
    assert e.read("commit")[0].tolist() == [3, 3, 3]
    assert e.read("applied")[0].tolist() == [3, 3, 3]
    assert e.read("vote")[0].tolist() == [0, 0, 0]
    assert e.read("term")[0].tolist() == [1, 1, 1]
    assert e.read("state")[0].tolist() == [2, 0, 0]
    assert e.start(0, 1, 400) == (-1, 1, False)          # not the leader: raft.go:345-351
    assert e.safety_check() == (0, 0)


def conflict_reply(mk):
    """raft.go:303-314: follower terms [0,1,1,2,2]; prev=4/prevTerm=3 -> ConflictTerm 2, first index 3;
    prev=7 -> ConflictTerm -1, ConflictFirstIndex len=5."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    _set_log(e, 0, 1, [0, 1, 1, 2, 2], term=3)
    assert e.append_entries(0, 1, 3, 0, 4, 3, [], 0) == (3, False, 2, 3)
    assert e.append_entries(0, 1, 3, 0, 7, 3, [], 0) == (3, False, -1, 5)
    assert e.read("loglen")[0, 1] == 5
    assert e.read("vote")[0, 1] == 0                      # raft.go:281
    # stale term: rejected with the receiver's term, nothing else touched (raft.go:276-278)
    assert e.append_entries(0, 1, 2, 2, 4, 2, [], 0) == (3, False, 0, 0)
    assert e.read("vote")[0, 1] == 0


def append_truncates_unconditionally(mk):
    """raft.go:291-292: on a match the log becomes logs[0:Prev+1] ++ Entries even if
    Entries is empty; commitIndex = min(LeaderCommit, len-1) (raft.go:293-299)."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    _set_log(e, 0, 2, [0, 1, 1, 2, 2], term=2, commit=1)
    assert e.append_entries(0, 2, 2, 0, 2, 1, [], 9) == (2, True, 0, 0)
    assert e.read("loglen")[0, 2] == 3
    assert e.read("commit")[0, 2] == 2
    assert e.append_entries(0, 2, 4, 1, 2, 1, [(3, 77), (4, 88)], 3) == (4, True, 0, 0)
    assert e.read("logterm")[0, 2, :5].tolist() == [0, 1, 1, 3, 4]
    assert e.read("logcmd")[0, 2, 3:5].tolist() == [77, 88]
    assert (e.read("term")[0, 2], e.read("vote")[0, 2], e.read("commit")[0, 2]) == (4, 1, 3)
    # persisted prefix is logs[0:commitIndex+1] only (raft.go:120)
    assert e.read("snaplen")[0, 2] == 4
    assert e.read("snaplogterm")[0, 2, :4].tolist() == [0, 1, 1, 3]


def leader_backoff(mk):
    """raft.go:493-505: leader terms [0,1,1,1,3,3], nextIndex 5; reply {ConflictTerm 2, first 3}
    -> scan 4->3, logs[3].Term=1 != 2 -> nextIndex 3; reply {1,1} -> stops at index 3 -> nextIndex 4.
    Driven through a real round: the follower's log makes it answer exactly those hints."""
    for fol_terms, want_next in (([0, 1, 1, 2, 2], 3), ([0, 1, 1, 1, 1], 4)):
        e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
        e.step(30)                                         # peer 0 leads term 1
        # the hints these follower logs produce for prev=4, prevTerm=3: {2,3} and {1,1}
        _set_log(e, 0, 0, [0, 1, 1, 1, 3, 3], term=3, commit=0)
        _set_log(e, 0, 1, fol_terms, term=3, commit=0)
        _set_log(e, 0, 2, fol_terms, term=3, commit=0)
        nx = e.read("nextindex")
        nx[0, 0, :] = 5
        e.write("nextindex", nx)
        hb = e.read("hbtimer")
        hb[0, 0] = -1000                                   # force a round on the next poll
        e.write("hbtimer", hb)
        e.step(2)
        assert e.read("nextindex")[0, 0, 1] == want_next, (fol_terms, e.read("nextindex")[0, 0])
        assert e.read("nextindex")[0, 0, 2] == want_next


def commit_rule(mk):
    """raft.go:400-416: P=5, len 8, all terms current, matchIndex of the others [7,5,3,0] -> commitIndex 5;
    with logs[5].Term < currentTerm the scan stops at the first older entry above it -> no commit."""
    for terms, want in (([0, 2, 2, 2, 2, 2, 2, 2], 5), ([0, 1, 1, 1, 1, 1, 2, 2], 0)):
        e = mk(n_groups=1, n_peers=5, log_cap=16, seed=1)
        e.step(30)                                         # peer 0 leads term 1
        for p in range(5):
            e.set_connected(0, p, p == 0)                  # rounds reach nobody: matchIndex stays as written
        _set_log(e, 0, 0, terms, term=2, commit=0)
        m = e.read("matchindex")
        m[0, 0, :] = [0, 7, 5, 3, 0]
        e.write("matchindex", m)
        e.step(3)
        assert e.read("commit")[0, 0] == want, terms
        if want:
            assert e.read("snaplen")[0, 0] == want + 1     # raft.go:413


def vote_compares_commit_index(mk):
    """raft.go:218-219: the up-to-date test uses commitIndex: voter commitIndex 4 with term 2 there;
    candidate (3,2) refused, (4,2) granted, (0,3) granted."""
    for lli, llt, want in ((3, 2, False), (4, 2, True), (0, 3, True)):
        e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
        _set_log(e, 0, 1, [0, 1, 1, 2, 2, 2], term=4, commit=4)
        assert e.request_vote(0, 1, 5, 2, lli, llt) == (5, want)
        assert e.read("term")[0, 1] == 5
        assert e.read("vote")[0, 1] == (2 if want else -1)
        assert e.read("state")[0, 1] == 0
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    _set_log(e, 0, 1, [0, 1], term=4, commit=1)
    assert e.request_vote(0, 1, 3, 2, 9, 9) == (4, False)  # stale term: raft.go:205-207
    assert e.read("term")[0, 1] == 4


def restore_repersists_one_entry(mk):
    """raft.go:144-148: readPersist calls persist() while commitIndex is still 0, so after a restart from
    a 6-entry blob: logLen 6, commitIndex 5, lastApplied 0, and the stored blob is cut to one entry."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.step(30)
    for cmd in (11, 12, 13, 14, 15):
        e.start(0, 0, cmd)
        e.step(12)
    assert e.read("snaplen")[0, 1] == 6
    blob = e.snapshot_export(0, 1)
    assert len(blob) == 16 + 16 * 6
    e.restart(0, 1)
    assert (e.read("loglen")[0, 1], e.read("commit")[0, 1], e.read("applied")[0, 1]) == (6, 5, 0)
    assert e.read("snaplen")[0, 1] == 1
    assert e.read("logcmd")[0, 1, 1:6].tolist() == [11, 12, 13, 14, 15]
    assert e.read("connected")[0, 1] == 0                  # start1 = crash1 first: config.go:137
    e.set_connected(0, 1, True)
    e.step(30)
    assert e.read("applied")[0, 1] == 5                    # replays the restored log from index 1
    assert e.read("snaplen")[0, 1] == 6


def start_on_isolated_leader(mk):
    """test_test.go:196-209: with 3 of 5 peers cut off the leader accepts Start at index 2 and
    nothing commits within 2 s."""
    e = mk(n_groups=1, n_peers=5, log_cap=16, seed=1)
    e.step(30)
    assert e.start(0, 0, 10)[0] == 1
    e.step(20)
    assert e.read("commit")[0].tolist() == [1] * 5
    for p in (1, 2, 3):
        e.set_connected(0, p, False)
    assert e.start(0, 0, 20) == (2, 1, True)
    e.step(200)
    assert e.read("commit")[0].max() == 1
    assert e.read("applied")[0].max() == 1


def stop_rule(mk):
    """raft.go:513-520 with GetState :97: after four rounds in which nobody answered the leader
    reports not-leader and refuses Start, but its state word stays 2."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.step(30)
    e.set_connected(0, 0, False)
    e.step(100)
    assert e.read("state")[0, 0] == 2 and e.read("stop")[0, 0] == 1 and e.read("cnt")[0, 0] == 4
    assert e.get_state(0, 0) == (1, False)
    assert e.start(0, 0, 5) == (-1, 1, False)


def applied_entry_overwritten(mk):
    """The reference elects on commitIndex (raft.go:218-219, 582-583) and persists only the committed
    prefix (raft.go:120), so an entry its leader committed and applied can be overwritten after that
    leader crashes.  5 peers, all [init,x1], commitIndex 1; A=0 leads term 1, holds x2 at index 2,
    counted it on B, C and applied it; B, C hold x2 but still have commitIndex 1.  A crashes; a peer
    of the rest wins term 2 with LastLogIndex 1 and its first entry truncates x2 on B, C: index 2 is
    applied twice with different commands and the state-machine flag must rise."""
    e = mk(n_groups=1, n_peers=5, log_cap=16, seed=1)
    e.step(30)
    assert e.start(0, 0, 111)[0] == 1
    e.step(20)
    assert e.read("applied")[0].tolist() == [1] * 5
    for p in (0, 1, 2):
        _set_log(e, 0, p, [0, 1, 1], cmds=[0, 111, 222])
    c, a = e.read("commit"), e.read("applied")
    c[0, 0], a[0, 0] = 2, 2
    e.write("commit", c)
    e.write("applied", a)
    rec, rm = e.read("applyrec"), e.read("recmax")
    rec[0, 2], rm[0] = 222, 2
    e.write("applyrec", rec)
    e.write("recmax", rm)
    e.crash(0, 0)
    tm = e.read("timer")
    tm[0, 3] = -5000                                       # D is the first to time out
    e.write("timer", tm)
    for _ in range(400):
        e.step(1)
        lead = [p for p in range(1, 5) if e.get_state(0, p)[1]]
        if lead:
            break
    assert lead, "no new leader"
    assert e.read("term")[0, lead[0]] == 2
    assert lead == [3]
    idx, _, ok = e.start(0, 3, 333)
    assert ok and idx == 2
    e.step(40)
    assert e.read("logcmd")[0, 1, 2] == 333 and e.read("logcmd")[0, 2, 2] == 333
    fl, n = e.safety_flags()
    assert fl & _abi.FLAG_STATE_MACHINE and n == 1


CASES = {f.__name__: f for f in (
    boot_and_three_starts, conflict_reply, append_truncates_unconditionally, leader_backoff, commit_rule,
    vote_compares_commit_index, restore_repersists_one_entry, start_on_isolated_leader, stop_rule,
    applied_entry_overwritten)}
