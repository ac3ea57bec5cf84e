"""The linearizability checker (include/lincheck.h) on hand-derived histories and against a brute-force
search.  The reference holds no test or fixture for src/porcupine: its only caller builds histories at run
time (src/kvraft/test_test.go:349-361), so the answers below are derived by hand from src/models/kv.go:40-54
and the definition of linearizability; agreement with the Go package on other inputs is parity unpinned."""
import itertools
import random

import pytest

from raft_amd import lincheck as lc
from raft_amd.lincheck import APPEND, GET, PUT, Event, Operation


def op(c, kind, key, value, output, call, ret):
    return Operation(c, kind, key, value, output, call, ret)


def test_library_exports():
    lib = lc.load_library()
    assert all(hasattr(lib, s) for s in lc.EXPORTS)


def test_bitset_hash_known_answers():
    # bitset.go:54-60: popcount xor-folded with every word
    assert lc.bitset_hash([0b1011]) == 3 ^ 0b1011
    assert lc.bitset_hash([1 << 63, 1]) == 2 ^ (1 << 63) ^ 1
    assert lc.bitset_hash([]) == 0


def test_sequential_history():
    h = [op(0, PUT, "x", "1", "", 0, 1), op(0, GET, "x", "", "1", 2, 3), op(1, APPEND, "x", "2", "", 4, 5), op(1, GET, "x", "", "12", 6, 7)]
    assert lc.check_operations(h)
    h[3] = op(1, GET, "x", "", "1", 6, 7)          # the append had returned before this get was called
    assert not lc.check_operations(h)


def test_stale_read_is_illegal_and_concurrent_read_is_not():
    stale = [op(0, PUT, "x", "a", "", 0, 10), op(1, PUT, "x", "b", "", 11, 20), op(2, GET, "x", "", "a", 21, 30)]
    assert not lc.check_operations(stale)
    overlap = [op(0, PUT, "x", "a", "", 0, 10), op(1, PUT, "x", "b", "", 11, 25), op(2, GET, "x", "", "a", 21, 30)]
    assert lc.check_operations(overlap)           # the get may take effect before the second put does


def test_concurrent_appends():
    base = [op(0, APPEND, "k", "a", "", 0, 10), op(1, APPEND, "k", "b", "", 0, 10)]
    assert lc.check_operations(base + [op(2, GET, "k", "", "ab", 11, 12)])
    assert lc.check_operations(base + [op(2, GET, "k", "", "ba", 11, 12)])
    assert not lc.check_operations(base + [op(2, GET, "k", "", "a", 11, 12)])
    # two reads that disagree on the order
    assert not lc.check_operations(base + [op(2, GET, "k", "", "ab", 11, 12), op(3, GET, "k", "", "ba", 13, 14)])


def test_keys_are_checked_apart():
    good = [op(0, PUT, "x", "1", "", 0, 1), op(1, PUT, "y", "2", "", 0, 1), op(0, GET, "x", "", "1", 2, 3), op(1, GET, "y", "", "2", 2, 3)]
    assert lc.check_operations(good)
    bad = good[:3] + [op(1, GET, "y", "", "1", 2, 3)]
    assert not lc.check_operations(bad)
    assert lc.check_operations([])                # the empty history


def test_initial_state_is_the_empty_string():
    assert lc.check_operations([op(0, GET, "x", "", "", 0, 1)])
    assert not lc.check_operations([op(0, GET, "x", "", "v", 0, 1)])


def test_events_interface():
    ev = [Event(0, False, 7, PUT, "x", "a"), Event(1, False, 9, GET, "x", ""), Event(0, True, 7, PUT, "x", ""), Event(1, True, 9, GET, "x", "a")]
    assert lc.check_events(ev)
    ev[3] = Event(1, True, 9, GET, "x", "b")
    assert not lc.check_events(ev)
    with pytest.raises(ValueError):
        lc.check_events([Event(0, True, 1, GET, "x", "")])       # a return without its call


def test_malformed_operations_are_rejected():
    with pytest.raises(ValueError):
        lc.check_operations([op(0, GET, "x", "", "", 5, 4)])
    with pytest.raises(ValueError):
        lc.check_operations([op(0, 9, "x", "", "", 0, 1)])


def test_timeout_reports_unknown():
    # 14 overlapping appends and a read no order can explain: the search has to visit every subset
    h = [op(i, APPEND, "k", chr(97 + i), "", 0, 100) for i in range(14)] + [op(99, GET, "k", "", "?", 101, 102)]
    assert lc.check_operations_timeout(h, 0.05) == lc.UNKNOWN
    small = h[:5] + [h[-1]]
    assert lc.check_operations_timeout(small, 5) == lc.ILLEGAL
    assert lc.last_cache_size() > 5


def _brute_force(history):
    """All orders that respect real time (a returns before b is called => a first), per key."""
    keys = sorted(set(o.key for o in history))
    for k in keys:
        ops = [o for o in history if o.key == k]
        found = False
        for perm in itertools.permutations(range(len(ops))):
            pos = {j: i for i, j in enumerate(perm)}
            if any(ops[a].ret < ops[b].call and pos[a] > pos[b] for a in range(len(ops)) for b in range(len(ops)) if a != b):
                continue
            state, ok = "", True
            for j in perm:
                o = ops[j]
                if o.op == GET:
                    if o.output != state:
                        ok = False
                        break
                elif o.op == PUT:
                    state = o.value
                else:
                    state += o.value
            if ok:
                found = True
                break
        if not found:
            return False
    return True


@pytest.mark.parametrize("seed", range(6))
def test_against_brute_force(seed):
    rng = random.Random(seed)
    agree_ok = agree_bad = 0
    for _ in range(150):
        n = rng.randint(1, 6)
        # distinct, odd/even separated times keep the sort of equal times out of the comparison
        times = rng.sample(range(100), 2 * n)
        h = []
        for i in range(n):
            a, b = sorted(times[2 * i:2 * i + 2])
            kind = rng.choice([GET, PUT, APPEND])
            key = rng.choice("xy")
            val = rng.choice("ab") if kind != GET else ""
            out = rng.choice(["", "a", "b", "ab", "ba", "aa"]) if kind == GET else ""
            h.append(op(i, kind, key, val, out, a, b))
        want = _brute_force(h)
        assert lc.check_operations(h) == want, h
        agree_ok += want
        agree_bad += not want
    assert agree_ok > 10 and agree_bad > 10
