#!/usr/bin/env python3
"""Replays one seed of tests/test_gpu_fuzz.py up to the first launch after which engine and oracle differ, then
(a) shortens that launch tick by tick to find the first tick at whose end they differ, and (b) cuts it in two at every
tick to see which cuts make the difference go away.  Seeds whose host verbs come before the difference are not handled.
usage: python profiles/fuzz_repro.py <seed>"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
from oracle.binding import OracleEngine
from tests.parity import diff_states
from tests import test_gpu_fuzz as F

seed = int(sys.argv[1])


def replay(upto_launch=None, cut=None, shorten=None):
    rng = random.Random(1000 + seed)
    kw = F._random_config(rng)
    o, e = OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))
    ops = None
    if rng.random() < 0.35:
        ops = F._random_ops(rng, kw["n_peers"], 1500)
        o.set_schedule(ops); e.set_schedule(ops)
    t, k = 0, 0
    while t < 1500:
        n = rng.choice([1, 1, 2, 3, 5, 8, 13, 50, 200, 600])
        if upto_launch == k and shorten is not None:
            n = shorten
        o.step(n)
        if upto_launch == k and cut is not None:
            if cut > 0: e.step(cut)
            if n - cut > 0: e.step(n - cut)
        else:
            e.step(n)
        t += n
        d = diff_states(o, e)
        if d or upto_launch == k:
            return kw, ops, o, e, k, t - n, n, d
        k += 1
        if rng.random() < 0.2:
            raise SystemExit("host verbs between launches before the difference: extend this script")
    return kw, ops, o, e, k, t, 0, []


kw, ops, o, e, k, t0, n, d = replay()
print("config", kw)
print("schedule", ops)
print("launch %d: ticks [%d, %d): %s" % (k, t0, t0 + n, d[:8]))
if d:
    for x in range(1, n + 1):
        _, _, o2, e2, _, _, _, d2 = replay(k, shorten=x)
        o2.close(); e2.close()
        if d2:
            print("first difference after a launch of %d ticks (tick %d): %s" % (x, t0 + x - 1, d2[:8]))
            break
    good = []
    for cut in range(0, n + 1):
        _, _, o2, e2, _, _, _, d2 = replay(k, cut)
        good.append(not d2)
        o2.close(); e2.close()
    print("cut -> equal:", "".join("1" if g else "0" for g in good))
