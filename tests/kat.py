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


# ---- the seeded election-timeout stream, restated from its definition (docs/LOCKSTEP.md, "Seeded streams") ----
_M = (1 << 64) - 1


def _sm64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M
    return x ^ (x >> 31)


def _timeout_draw(seed, group, peer, draw):
    """nextRandom() (raft.go:532-534) on the peer's stream: 500 + bits 32..63 of hash(key, 1, peer, draw) % 600."""
    key = _sm64(_sm64(seed) ^ group)
    h = _sm64(key ^ (1 << 56) ^ (peer << 32) ^ draw)
    return 500 + (h >> 32) % 600


def _put(e, field, g, p, v):
    a = e.read(field)
    a[g, p] = v
    e.write(field, a)


def same_term_vote_request_demotes_leader(mk):
    """raft.go:208-216: a RequestVote whose term is not below the receiver's sets state = 0 before the vote test,
    so a leader is demoted by a candidate of its OWN term, although it refuses the vote (it voted for itself)."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.step(30)
    assert e.get_state(0, 0) == (1, True)
    assert e.request_vote(0, 0, 1, 2, 0, 0) == (1, False)   # same term 1, candidate 2
    assert (e.read("state")[0, 0], e.read("term")[0, 0], e.read("vote")[0, 0]) == (0, 1, 0)
    assert e.get_state(0, 0) == (1, False)
    assert e.start(0, 0, 9) == (-1, 1, False)
    assert e.read("monalive")[0, 0] == 1                     # the monitor goroutine notices at its next poll
    e.step(3)
    assert e.read("monalive")[0, 0] == 0                     # raft.go:396,526-528
    assert e.read("state")[0].tolist() == [0, 0, 0]


def candidate_steps_down_without_adopting_term(mk):
    """raft.go:618-627: a candidate that sees a higher term in a reply becomes follower with voteFor = -1 but
    keeps its own currentTerm, ends the round (the peers after it are not called) and draws a new timeout
    (raft.go:639).  Peer 1 is a candidate at term 3 -> round at term 4; peer 0 is at term 7; peer 2 is never asked."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    _put(e, "term", 0, 0, 7)
    _put(e, "term", 0, 1, 3)
    _put(e, "term", 0, 2, 3)
    _put(e, "state", 0, 1, 1)
    e.step(1)
    c = e.counters()
    assert (c["rv_rounds"], c["rpc_sent"], c["rpc_replied"], c["stepdowns"], c["elections_won"]) == (1, 1, 1, 1, 0)
    assert e.read("term")[0].tolist() == [7, 4, 3]           # 3 + 1 (raft.go:570); 7 is NOT adopted
    assert e.read("vote")[0].tolist() == [-1, -1, -1]        # voteFor = me (:575), then -1 (:624); peer 2 untouched
    assert e.read("state")[0].tolist() == [0, 0, 0]
    assert (e.read("snapterm")[0, 1], e.read("snapvote")[0, 1]) == (4, -1)   # persist() at :625
    assert e.read("restart")[0, 1] == 0
    assert e.read("rngctr")[0, 1] == 2                       # Make (:546) and the draw at :639
    assert e.read("timeout")[0, 1] == _timeout_draw(1, 0, 1, 1)
    assert e.read("timer")[0].tolist() == [0, 0, 0]          # the stale-term handler call resets nothing (:205-207)


def leader_adopts_term_keeps_vote(mk):
    """raft.go:479-487: a leader that sees a higher term in an AppendEntries reply becomes follower AT that term,
    persists, and ends the round; voteFor keeps its old value (itself).  Peer 1 is at term 5, so peer 2 is not
    called in that round."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.step(30)
    assert e.get_state(0, 0) == (1, True) and e.read("vote")[0, 0] == 0
    _put(e, "term", 0, 1, 5)
    _put(e, "hbtimer", 0, 0, -1000)                          # the gate of raft.go:421 opens at the next poll
    before = e.counters()
    e.step(2)
    c = e.counters()
    assert c["ae_rounds"] - before["ae_rounds"] == 1
    assert c["rpc_sent"] - before["rpc_sent"] == 1           # break at :487: peer 2 is not called
    assert c["stepdowns"] - before["stepdowns"] == 1
    assert (e.read("term")[0, 0], e.read("vote")[0, 0], e.read("state")[0, 0]) == (5, 0, 0)
    assert (e.read("snapterm")[0, 0], e.read("snapvote")[0, 0]) == (5, 0)
    assert e.read("term")[0, 2] == 1
    assert e.read("monalive")[0, 0] == 0                     # next loop iteration: not leader, exit (:526-528)


def votes_are_counted_per_reply(mk):
    """raft.go:585-616: all RequestVote RPCs of a round decode into ONE reply struct and gob leaves a field
    alone when the wire value is zero (labgob.go:116-121), so VoteGranted is cleared by hand after it is
    counted (:615).  One grant followed by three refusals is therefore 2 votes of 5, not 5: no leader, and the
    candidate is marked to retry (restart, :660)."""
    e = mk(n_groups=1, n_peers=5, log_cap=16, seed=1)
    _put(e, "state", 0, 0, 1)                                # candidate; round at term 1
    for p in (2, 3, 4):                                      # already voted for peer 4 in term 1
        _put(e, "term", 0, p, 1)
        _put(e, "vote", 0, p, 4)
    e.step(1)
    c = e.counters()
    assert (c["rv_rounds"], c["rpc_sent"], c["rpc_delivered"], c["rpc_replied"], c["elections_won"]) == (1, 4, 4, 4, 0)
    assert e.read("vote")[0].tolist() == [0, 0, 4, 4, 4]
    assert e.read("term")[0].tolist() == [1, 1, 1, 1, 1]
    assert (e.read("state")[0, 0], e.read("restart")[0, 0], e.read("timeout")[0, 0]) == (1, 1, 1000)
    assert e.read("state")[0, 4] == 0                        # raft.go:216


def late_vote_replies_are_lost(mk):
    """raft.go:605-610: a reply that comes after the 25 ms wait is never looked at (in the reference it would
    land in the shared struct later: a data race).  With every reply long-delayed (labrpc.go:276-285) all four
    voters grant -- their handlers ran -- and the candidate counts only itself; it waited 4 x 25 ms."""
    e = mk(n_groups=1, n_peers=5, log_cap=16, seed=1, reply_delay_permille=1000)
    _put(e, "state", 0, 0, 1)
    e.step(1)
    c = e.counters()
    assert (c["rpc_sent"], c["rpc_delivered"], c["rpc_replied"], c["elections_won"]) == (4, 4, 0, 0)
    assert e.read("vote")[0].tolist() == [0, 0, 0, 0, 0]
    assert (e.read("state")[0, 0], e.read("restart")[0, 0]) == (1, 1)
    assert 150 <= e.read("elnext")[0, 0] <= 153              # 4 x 25 ms of waits, then the 50 ms sleep (+ slack)


def failed_candidate_retries_same_term(mk):
    """raft.go:569-573, 660: a candidate that neither won nor stepped down sets restart and its next round does
    NOT increment currentTerm.  Peer 0 (timeout 0 at boot, raft.go:545) is cut off: it turns candidate at its
    second loop iteration, runs a round every 100 ms + 50 ms sleep, and stays at term 1 for ever; the round
    sets timeout = 1000 (:567) without a draw."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.set_connected(0, 0, False)
    e.step(8)
    assert (e.read("state")[0, 0], e.read("term")[0, 0], e.read("rngctr")[0, 0]) == (1, 0, 1)   # :557-561, one draw
    e.step(12)
    assert (e.read("term")[0, 0], e.read("restart")[0, 0], e.read("roundseq")[0, 0]) == (1, 1, 1)
    e.step(180)
    rounds = e.read("roundseq")[0, 0]
    assert rounds >= 10
    assert (e.read("term")[0, 0], e.read("state")[0, 0], e.read("restart")[0, 0]) == (1, 1, 1)
    assert (e.read("timeout")[0, 0], e.read("rngctr")[0, 0]) == (1000, 1)
    assert e.read("vote")[0, 0] == 0
    return e


def rng_draw_points(mk):
    """nextRandom() is called at Make unless (term 0, peer 0) (raft.go:545-547), on follower -> candidate (:561)
    and after a round that ended in a step-down (:639); the round itself sets timeout = 1000 without a draw
    (:567), winning draws nothing, and a candidate demoted by a HANDLER keeps timeout 1000 (the loop's local
    variable is only redrawn at :561/:639)."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    assert e.read("rngctr")[0].tolist() == [0, 1, 1]
    assert e.read("timeout")[0].tolist() == [0, _timeout_draw(1, 0, 1, 0), _timeout_draw(1, 0, 2, 0)]
    e.step(30)                                               # peer 0: candidate (draw), round (no draw), wins (no draw)
    assert e.get_state(0, 0) == (1, True)
    assert e.read("rngctr")[0].tolist() == [1, 1, 1]
    assert e.read("timeout")[0].tolist() == [1000, _timeout_draw(1, 0, 1, 0), _timeout_draw(1, 0, 2, 0)]
    e.restart(0, 1)                                          # Make at term 1: one draw (:546)
    assert e.read("rngctr")[0, 1] == 2 and e.read("timeout")[0, 1] == _timeout_draw(1, 0, 1, 1)
    # a candidate demoted by AppendEntries (raft.go:280): no draw, timeout stays 1000
    e = failed_candidate_retries_same_term(mk)
    e.set_connected(0, 0, True)
    for _ in range(300):
        e.step(1)
        if e.read("state")[0, 0] == 0:
            break
    assert e.read("state")[0, 0] == 0
    assert (e.read("timeout")[0, 0], e.read("rngctr")[0, 0]) == (1000, 1)


def apply_stream_keeps_what_was_applied(mk):
    """applyCh carries logs[lastApplied].Command as it is when the entry is applied (raft.go:677-685), and a
    restarted peer replays from index 1 (lastApplied = 0 at Make, raft.go:717).  Same schedule as
    applied_entry_overwritten, but peer 0 applies x2 = 222 at index 2 itself, crashes, and comes back after the
    others replaced index 2 by 333: drained only at the very end, its stream is still (2, 222), then the replay
    (1, 111), then (2, 333) -- not what its log holds at drain time."""
    e = mk(n_groups=1, n_peers=5, log_cap=16, seed=1, apply_ring_cap=16)
    e.step(30)
    assert e.start(0, 0, 111)[0] == 1
    e.step(20)
    assert [e.drain_applied(0, p) for p in range(5)] == [[(1, 111)]] * 5
    for p in (0, 1, 2):
        _set_log(e, 0, p, [0, 1, 1], cmds=[0, 111, 222])
    e.set_connected(0, 0, False)
    _put(e, "commit", 0, 0, 2)                             # the leader counted x2 on B, C (raft.go:409-411) ...
    e.step(1)
    assert e.read("applied")[0, 0] == 2                    # ... and applied it
    e.crash(0, 0)
    _put(e, "timer", 0, 3, -5000)                          # D is the first to time out
    for _ in range(400):
        e.step(1)
        if e.get_state(0, 3)[1]:
            break
    assert e.get_state(0, 3) == (2, True)
    assert e.start(0, 3, 333) == (2, 2, True)
    e.step(40)
    e.restart(0, 0)                                        # from its persisted [init, 111] (raft.go:120: commitIndex was 1 at its last persist)
    e.set_connected(0, 0, True)
    e.step(60)
    assert e.read("logcmd")[0, :, 2].tolist() == [333] * 5
    assert e.drain_applied(0, 0) == [(2, 222), (1, 111), (2, 333)]
    assert [e.drain_applied(0, p) for p in (1, 2, 3, 4)] == [[(2, 333)]] * 4
    assert e.drain_applied(0, 0) == []
    assert e.safety_flags()[0] & _abi.FLAG_STATE_MACHINE


def apply_ring_overrun_is_reported(mk):
    """A reader that lets more than apply_ring_cap messages pile up is told so once (RG_ERR_OVERRUN) and gets
    the newest apply_ring_cap messages; without a ring there is no applyCh to read (RG_ERR_STATE)."""
    from raft_amd.engine import RaftGpuError
    e = mk(n_groups=1, n_peers=3, log_cap=32, seed=1, apply_ring_cap=4)
    e.step(30)
    for i in range(10):
        assert e.start(0, 0, 100 + i)[0] == i + 1
    e.step(30)
    assert e.read("applied")[0].tolist() == [10, 10, 10]
    try:
        e.drain_applied(0, 1)
        raise AssertionError("overrun not reported")
    except RaftGpuError as err:
        assert err.status == _abi.ERR_OVERRUN
        assert err.partial == [(7, 106), (8, 107), (9, 108), (10, 109)]
    assert e.drain_applied(0, 1) == []
    assert e.start(0, 0, 555)[0] == 11
    e.step(30)
    assert e.drain_applied(0, 1) == [(11, 555)]
    e = mk(n_groups=1, n_peers=3, log_cap=32, seed=1)
    try:
        e.drain_applied(0, 0)
        raise AssertionError("drain without a ring")
    except RaftGpuError as err:
        assert err.status == _abi.ERR_STATE


def vote_test_reads_the_voters_own_entry(mk):
    """raft.go:218: the voter compares with logs[commitIndex].Term of ITS OWN log.  This reference elects on
    commitIndex, so two logs that agree at PrevLogIndex may differ below it; when a heartbeat only raises the
    follower's commitIndex (raft.go:293-299, no entry carried) the term at the new commitIndex is the follower's,
    not the leader's.  Leader terms [0,1,2,3,3], followers [0,1,1,3,3], PrevLogIndex 4 matches, LeaderCommit 2:
    the followers then grant a candidate with (LastLogIndex 2, LastLogTerm 1) -- 1 >= their own term 1 at
    index 2, though the leader holds term 2 there."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.step(30)
    assert e.get_state(0, 0) == (1, True)
    _set_log(e, 0, 0, [0, 1, 2, 3, 3], term=4, commit=2)      # no entry of term 4: the commit rule stays put (raft.go:402)
    for p in (1, 2):
        _set_log(e, 0, p, [0, 1, 1, 3, 3], commit=1)
    nx = e.read("nextindex")
    nx[0, 0, :] = 5
    e.write("nextindex", nx)
    _put(e, "hbtimer", 0, 0, -1000)
    e.step(2)
    assert e.read("commit")[0].tolist() == [2, 2, 2] and e.read("loglen")[0].tolist() == [5, 5, 5]
    assert e.read("term")[0].tolist() == [4, 4, 4]
    assert e.request_vote(0, 1, 5, 2, 2, 1) == (5, True)
    assert e.request_vote(0, 2, 5, 1, 1, 1) == (5, False)     # LastLogIndex 1 < commitIndex 2 at equal terms


def _net_call(seed, group, s, j, seq, req_drop, reply_drop, reply_delay, reachable=True):
    """One Call on an unreliable net, restated from docs/LOCKSTEP.md "Network": (what happens, ms the sender waits)."""
    key = _sm64(_sm64(seed) ^ group)
    h = _sm64(key ^ (2 << 56) ^ (((s << 8) | j) << 32) ^ seq)
    if not reachable:
        return "unreachable", min(((h >> 32) * 7000) >> 32, 25)   # labrpc.go:290-305: Call returns false after 0..6999 ms
    d = (((h >> 48) & 0xFF) * 27) >> 8                       # labrpc.go:222: 0..26 ms
    permille = lambda x: (x * 1000) >> 16
    if permille(h & 0xFFFF) < req_drop:
        return "lost", min(d, 25)                            # labrpc.go:226-229: Call returns false after the short delay
    if permille((h >> 16) & 0xFFFF) < reply_drop:
        return "reply lost", min(d, 25)                      # labrpc.go:273-275
    if d >= 25 or permille((h >> 32) & 0xFFFF) < reply_delay:
        return "late", 25                                    # raft.go:473: the sender has given up
    return "round trip", d


def a_failed_call_costs_its_short_delay(mk):
    """labrpc.go:222-229, 273-275 with raft.go:469-474, 521: the leader's select waits for each Call until it returns --
    false right after the net's 0..26 ms delay when the net drops the request or the reply -- or 25 ms, and
    heartbeatTimer is set when the round's Calls are through.  Twelve rounds of a leader on a net that drops 30 % of the
    requests and 30 % of the replies, every Call worked out here from the hash alone: what reached whom, and how long
    the round took."""
    seed = 1
    e = mk(n_groups=1, n_peers=3, log_cap=64, seed=seed)
    e.step(30)
    assert e.get_state(0, 0) == (1, True)
    e.set_unreliable(300, 300, 0)
    kinds, rounds = set(), 0
    for _ in range(400):
        if rounds == 12:
            break
        assert e.get_state(0, 0)[1]
        q = int(e.read("roundseq")[0, 0])
        e.start(0, 0, 1000 + e.now())
        timers, match = e.read("timer")[0].copy(), e.read("matchindex")[0, 0].copy()
        e.step(1)
        if int(e.read("roundseq")[0, 0]) == q:
            continue
        assert int(e.read("roundseq")[0, 0]) == q + 1
        now, cost = (e.now() - 1) * 10, 0
        length = int(e.read("loglen")[0, 0])
        for j in (1, 2):
            kind, ms = _net_call(seed, 0, 0, j, q, 300, 300, 0)
            kinds.add(kind)
            cost += ms
            if kind == "lost":
                assert e.read("timer")[0, j] == timers[j]                       # the handler never ran
            else:
                assert e.read("timer")[0, j] == now and e.read("loglen")[0, j] == length
            if kind == "round trip":
                assert e.read("matchindex")[0, 0, j] == length - 1
            else:
                assert e.read("matchindex")[0, 0, j] == match[j]                # the leader heard nothing
        assert e.read("hbtimer")[0, 0] == now + cost
        rounds += 1
    assert rounds == 12 and kinds == {"lost", "reply lost", "late", "round trip"}
    return e


def partitions_and_commands_follow_their_hashes(mk):
    """The device-generated workload, restated from docs/LOCKSTEP.md: a reshuffle takes ONE hash (stream 4, the
    group's reshuffle number) -- bit p is peer p's side, the upper half scaled to the jitter gives the next reshuffle's
    tick -- and the client's n-th command is hash(key, 3, 0, n).  Group 2 of 4, so the key is not group 0's."""
    seed, g, period, jitter = 3, 2, 50, 20
    e = mk(n_groups=4, n_peers=5, log_cap=64, seed=seed, part_period=period, part_jitter=jitter, start_period=7, start_first_tick=3, start_max=20)
    key = _sm64(_sm64(seed) ^ g)
    assert (int(e.read("partnext")[g]), int(e.read("partepoch")[g])) == (period, 0)
    for _ in range(6):
        at, epoch = int(e.read("partnext")[g]), int(e.read("partepoch")[g])
        e.step(at + 1 - e.now())
        h = _sm64(key ^ (4 << 56) ^ (epoch << 32))
        assert e.read("side")[g].tolist() == [(h >> p) & 1 for p in range(5)]
        assert int(e.read("partnext")[g]) == at + period + (((h >> 32) * jitter) >> 32)
        assert int(e.read("partepoch")[g]) == epoch + 1
    e.step(400)
    assert int(e.read("startseq")[g]) == 20
    cmds = [_sm64(key ^ (3 << 56) ^ n) for n in range(20)]
    have = e.read("logcmd")[g]
    lens = e.read("loglen")[g]
    assert int(lens.max()) == 21
    for p in range(5):
        assert [int(x) for x in have[p][1:int(lens[p])]] == cmds[:int(lens[p]) - 1]   # every log is a prefix of the command sequence
    return e


def churn_follows_its_hashes(mk):
    """test_test.go:893-919 as docs/LOCKSTEP.md fixes it: every churn_period ticks three hashes decide, in this order,
    whether a peer is disconnected, whether one is restarted (if dead) and connected, whether one is crashed.  The
    connectivity and liveness of a group replayed here from the hashes alone over 39 churn ticks."""
    seed, g, P, period = 5, 1, 5, 20
    rates = (300, 600, 300)
    e = mk(n_groups=3, n_peers=P, log_cap=64, seed=seed, churn_period=period, churn_disconnect_permille=rates[0],
           churn_restart_permille=rates[1], churn_crash_permille=rates[2])
    key = _sm64(_sm64(seed) ^ g)
    conn, dead, seen = [1] * P, [0] * P, set()
    for it in range(1, 40):
        e.step(it * period + 1 - e.now())
        hs = [_sm64(key ^ (5 << 56) ^ (it << 32) ^ k) for k in range(3)]
        hit = [(h & 0xFFFFF) % 1000 < r for h, r in zip(hs, rates)]
        who = [(h >> 32) % P for h in hs]
        if hit[0]:
            conn[who[0]] = 0
            seen.add("disconnect")
        if hit[1]:
            if dead[who[1]]:
                seen.add("restart")
            dead[who[1]], conn[who[1]] = 0, 1
        if hit[2] and not dead[who[2]]:
            conn[who[2]], dead[who[2]] = 0, 1
            seen.add("crash")
        assert e.read("connected")[g].tolist() == conn and e.read("dead")[g].tolist() == dead, "churn tick %d" % it
    assert seen == {"disconnect", "restart", "crash"}
    return e


def election_loop_sleeps_with_its_slack(mk):
    """raft.go:548-668 under docs/LOCKSTEP.md "Time": an idle follower's election loop wakes, finds its timer not run
    out, and sleeps 50 ms plus a seeded slack of 0..3 ms -- the 32-bit mix of (key, peer, now) restated here.  The
    wake-up times of two followers over 120 ticks, one tick per launch, so nothing is hidden by catching up."""
    m32 = 0xFFFFFFFF

    def slack(key, p, now):
        h = ((key & m32) ^ ((p * 0x9E3779B1) & m32) ^ ((now * 0x85EBCA6B) & m32)) & m32
        h ^= h >> 16
        h = (h * 0x7FEB352D) & m32
        h ^= h >> 15
        h = (h * 0x846CA68B) & m32
        h ^= h >> 16
        return h >> 30

    seed, g = 9, 1
    e = mk(n_groups=2, n_peers=3, log_cap=16, seed=seed)
    key = _sm64(_sm64(seed) ^ g)
    el, seen = e.read("elnext")[g].tolist(), set()
    assert el == [0, 0, 0]
    for t in range(120):
        e.step(1)
        for p in (1, 2):
            if el[p] <= 10 * t:
                seen.add(slack(key, p, 10 * t))
                el[p] = 10 * t + 50 + slack(key, p, 10 * t)
        assert e.read("elnext")[g].tolist()[1:] == el[1:], "tick %d" % t
    assert seen == {0, 1, 2, 3} and e.get_state(g, 0) == (1, True)
    return e


def candidate_rounds_on_a_lossy_net(mk):
    """raft.go:566-660 with labrpc.go:216-307: six RequestVote rounds of one candidate on a net that drops 40 % of the
    requests and 40 % of the replies, every Call worked out from its hash: a peer adopts the candidate's term exactly
    when the request reached it, a failed round ends in a sleep of 50 ms + slack after the round's waits, and the round
    that wins sets heartbeatTimer to the end of its waits."""
    m32 = 0xFFFFFFFF

    def slack(key, p, now):
        h = ((key & m32) ^ ((p * 0x9E3779B1) & m32) ^ ((now * 0x85EBCA6B) & m32)) & m32
        h ^= h >> 16
        h = (h * 0x7FEB352D) & m32
        h ^= h >> 15
        h = (h * 0x846CA68B) & m32
        h ^= h >> 16
        return h >> 30

    seed = 15
    e = mk(n_groups=1, n_peers=5, log_cap=16, seed=seed, req_drop_permille=400, reply_drop_permille=400)
    key = _sm64(_sm64(seed))
    rounds = 0
    for t in range(200):
        q, state, terms = int(e.read("roundseq")[0, 0]), int(e.read("state")[0, 0]), e.read("term")[0].copy()
        e.step(1)
        if int(e.read("roundseq")[0, 0]) == q:
            continue
        assert state == 1 and int(e.read("roundseq")[0, 0]) == q + 1      # a candidate's round (peer 0: timeout 0 at boot, raft.go:545)
        now, cost, term = 10 * t, 0, int(e.read("term")[0, 0])
        for j in range(1, 5):
            kind, ms = _net_call(seed, 0, 0, j, q, 400, 400, 0)
            cost += ms
            assert int(e.read("term")[0, j]) == (term if kind != "lost" else int(terms[j]))
        rounds += 1
        if int(e.read("state")[0, 0]) == 2:
            assert int(e.read("hbtimer")[0, 0]) == now + cost                       # raft.go:645-651
            break
        assert int(e.read("elnext")[0, 0]) == now + cost + 50 + slack(key, 0, now)  # raft.go:660 after the round's waits
        assert int(e.read("term")[0, 0]) == 1                                       # a failed round is retried in the same term
    assert rounds == 6 and e.get_state(0, 0) == (1, True)
    return e


def kill_leaves_the_peer_on_the_net(mk):
    """rf.Kill() (raft.go:377-380) only sets the dead flag: the peer's handlers return at once (:203, :273) but labrpc
    still delivers to its server, so Call returns true with the reply as the caller made it.  A leader therefore counts
    a killed follower active (:476-477), reads {Term 0, Success false, ConflictTerm 0, ConflictFirstIndex 0} (:456-461)
    and walks nextIndex down to 1 (:493-499: logs[0].Term == 0 == ConflictTerm); its silent-round counter never moves
    (:514), unlike with a crashed follower (stop_rule).  A candidate counts killed peers active without their vote."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.step(30)
    for cmd in (7, 8, 9):
        assert e.start(0, 0, cmd)[2]
        e.step(12)
    assert e.read("nextindex")[0, 0].tolist()[1:] == [4, 4] and e.read("commit")[0].tolist() == [3, 3, 3]
    e.kill(0, 2)
    assert (e.read("dead")[0].tolist(), e.read("connected")[0].tolist()) == ([0, 0, 2], [1, 1, 1])
    assert e.get_state(0, 2) == (1, False) and e.start(0, 2, 1) == (-1, 1, False)
    before = e.counters()
    timer2 = int(e.read("timer")[0, 2])
    e.step(11)                                             # one heartbeat round (100 ms gate, raft.go:421)
    c = e.counters()
    assert c["ae_rounds"] - before["ae_rounds"] == 1
    assert (c["rpc_sent"] - before["rpc_sent"], c["rpc_delivered"] - before["rpc_delivered"], c["rpc_replied"] - before["rpc_replied"]) == (2, 2, 2)
    assert c["ae_conflicts"] == before["ae_conflicts"]     # the handler never reached :303
    assert e.read("nextindex")[0, 0].tolist()[1:] == [4, 1]
    assert e.read("matchindex")[0, 0, 2] == 3              # untouched
    assert int(e.read("timer")[0, 2]) == timer2 and e.read("loglen")[0, 2] == 4
    e.kill(0, 1)
    e.step(100)                                            # ten more rounds that "reach" everybody: no stop (raft.go:514)
    assert (e.read("cnt")[0, 0], e.read("stop")[0, 0]) == (0, 0) and e.get_state(0, 0) == (1, True)
    assert e.read("nextindex")[0, 0].tolist()[1:] == [1, 1]
    assert e.start(0, 0, 10) == (4, 1, True)
    e.step(30)
    assert e.read("commit")[0].tolist() == [3, 3, 3]       # nothing commits without a live majority
    e.crash(0, 1)                                          # crash1 deletes the server (config.go:99-127): now unreachable
    assert (e.read("dead")[0, 1], e.read("connected")[0, 1]) == (1, 0)
    e.restart(0, 2)                                        # start1: a fresh peer from the Persister
    assert (e.read("dead")[0, 2], e.read("connected")[0, 2], e.read("loglen")[0, 2]) == (0, 0, 4)
    # a candidate among killed peers: both Calls come back true, neither carries a vote
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.kill(0, 1)
    e.kill(0, 2)
    _put(e, "state", 0, 0, 1)
    e.step(1)
    c = e.counters()
    assert (c["rv_rounds"], c["rpc_sent"], c["rpc_delivered"], c["rpc_replied"], c["elections_won"]) == (1, 2, 2, 2, 0)
    assert e.read("term")[0].tolist() == [1, 0, 0] and e.read("vote")[0].tolist() == [0, -1, -1]
    assert (e.read("state")[0, 0], e.read("restart")[0, 0]) == (1, 1)
    assert 50 <= e.read("elnext")[0, 0] <= 53              # no waits: both Calls returned at once on the reliable net


def two_leaders_in_one_term(mk):
    """The election loop's `restart` (raft.go:544) outlives a term change made by a handler: a candidate whose round
    failed at term T sets restart (:660); a RequestVote of term T+1 then makes it a follower of T+1 and takes its vote
    (:209-222); when it times out, its next round does NOT increment currentTerm (:569-573) but sets voteFor = me (:575):
    a second vote in term T+1, and a second leader of T+1.  Five peers A..E = 0..4, partitions by side (both leaders
    stay `connected`, which is what the tester's checkOneLeader looks at, config.go:307-320)."""
    e = mk(n_groups=1, n_peers=5, log_cap=16, seed=1)
    for p in range(5):
        _put(e, "timer", 0, p, 1000000)                    # nobody times out unless told to
    for p in (1, 2, 3, 4):
        e.set_side(0, p, 1)                                # A alone
    _put(e, "state", 0, 0, 1)
    e.step(1)                                              # A's round at term 1 reaches nobody
    assert (e.read("term")[0, 0], e.read("vote")[0, 0], e.read("restart")[0, 0], e.read("state")[0, 0]) == (1, 0, 1, 1)
    assert e.read("elnext")[0, 0] >= 100                   # its retry lies beyond the next tick
    for p, side in enumerate((0, 0, 0, 1, 1)):             # {A, B, C} | {D, E}
        e.set_side(0, p, side)
    _put(e, "term", 0, 1, 1)
    _put(e, "state", 0, 1, 1)
    _put(e, "elnext", 0, 1, 10)
    e.step(1)                                              # B's round at term 2: A and C grant
    assert e.get_state(0, 1) == (2, True)
    assert e.read("term")[0].tolist() == [2, 2, 2, 0, 0] and e.read("vote")[0].tolist() == [1, 1, 1, -1, -1]
    assert (e.read("state")[0, 0], e.read("restart")[0, 0]) == (0, 1)          # follower of term 2, restart still set
    assert e.safety_flags() == (0, 0)
    for p, side in enumerate((0, 1, 1, 0, 0)):             # {B, C} | {A, D, E}: B keeps a follower, so it never stops (:514)
        e.set_side(0, p, side)
    _put(e, "timer", 0, 0, -5000)                          # A times out at its next loop iteration
    for _ in range(40):
        e.step(1)
        if e.get_state(0, 0)[1]:
            break
    assert e.get_state(0, 0) == (2, True) and e.get_state(0, 1) == (2, True)   # two leaders of term 2
    assert e.read("vote")[0].tolist() == [0, 1, 1, 0, 0]                       # A voted for B and for itself in term 2
    assert e.read("connected")[0].tolist() == [1] * 5
    fl, n = e.safety_flags()
    assert fl & _abi.FLAG_ELECTION_SAFETY and n == 1
    return e


def commit_index_past_a_truncated_log(mk):
    """The vote test compares commitIndex and the term there (raft.go:218-219), so a candidate whose whole log is shorter
    than a voter's commitIndex can win that vote with a newer term at its own commitIndex.  As leader it backs
    nextIndex off (:493-505) until PrevLogIndex matches, and the voter's log becomes logs[0:Prev+1] ++ Entries whatever
    its length was (:291-292); commitIndex is never lowered (:293-299), so commitIndex > len(logs)-1 and persist()
    re-slices logs[0:commitIndex+1] past the end (:120: the stale entries of the backing array; apply() at :679 would
    panic): RG_FLAG_REF_PANIC.  X = peer 1: log terms [0,3], commitIndex 1; Y = peer 2: [0,1,1,2,2], commitIndex 4."""
    e = mk(n_groups=1, n_peers=3, log_cap=16, seed=1)
    e.crash(0, 0)
    _set_log(e, 0, 1, [0, 3], cmds=[0, 31], term=3, commit=1)
    _set_log(e, 0, 2, [0, 1, 1, 2, 2], cmds=[0, 11, 12, 21, 22], term=3, commit=4)
    _put(e, "applied", 0, 1, 1)
    _put(e, "applied", 0, 2, 4)
    _put(e, "state", 0, 1, 1)
    e.step(1)                                              # X's round at term 4: {4, X, LastLogIndex 1, LastLogTerm 3} beats (4, term 2)
    assert e.get_state(0, 1) == (4, True) and e.read("vote")[0, 2] == 1
    assert e.read("nextindex")[0, 1, 2] == 2               # len(logs) of the new leader (:649-652)
    seen_backoff = False
    for _ in range(40):
        e.step(1)
        seen_backoff |= int(e.read("nextindex")[0, 1, 2]) == 1    # {ConflictTerm 1, first index 1}; the leader has no term-1 entry
        if e.read("loglen")[0, 2] == 2:
            break
    assert seen_backoff
    assert (e.read("loglen")[0, 2], e.read("commit")[0, 2]) == (2, 4)          # truncated below commitIndex
    assert e.read("logterm")[0, 2, :2].tolist() == [0, 3] and e.read("logcmd")[0, 2, 1] == 31
    assert e.read("snaplen")[0, 2] == 5                    # persist() at :300 took logs[0:5]
    assert e.read("snaplogterm")[0, 2, :5].tolist() == [0, 3, 1, 2, 2]         # ... the new entry and the stale words behind it
    fl, n = e.safety_flags()
    assert fl & _abi.FLAG_REF_PANIC and n == 1
    return e


CASES = {f.__name__: f for f in (
    boot_and_three_starts, conflict_reply, append_truncates_unconditionally, leader_backoff, commit_rule,
    vote_compares_commit_index, restore_repersists_one_entry, start_on_isolated_leader, stop_rule,
    applied_entry_overwritten, same_term_vote_request_demotes_leader, candidate_steps_down_without_adopting_term,
    leader_adopts_term_keeps_vote, votes_are_counted_per_reply, late_vote_replies_are_lost,
    failed_candidate_retries_same_term, rng_draw_points, apply_stream_keeps_what_was_applied,
    apply_ring_overrun_is_reported, vote_test_reads_the_voters_own_entry, a_failed_call_costs_its_short_delay,
    partitions_and_commands_follow_their_hashes, churn_follows_its_hashes,
    election_loop_sleeps_with_its_slack, candidate_rounds_on_a_lossy_net, kill_leaves_the_peer_on_the_net,
    two_leaders_in_one_term, commit_index_past_a_truncated_log)}
