"""Randomised schedules and launch lengths, HIP engine vs CPU oracle (-m gpu).  The engine visits each group's
ticks in its own order and defers work that changes nothing yet; whatever the launch boundaries, every word has
to match the tick-by-tick oracle when a launch ends."""
import random

import numpy as np
import pytest

from raft_amd import _abi
from tests.parity import diff_states

pytestmark = pytest.mark.gpu


def _random_config(rng):
    peers = rng.choice([2, 3, 3, 4, 5, 5, 5, 6, 7, 8])
    kw = dict(n_groups=rng.choice([1, 7, 13, 40]), n_peers=peers, log_cap=rng.choice([32, 96, 300]), seed=rng.randrange(1 << 30))
    if rng.random() < 0.7:
        d = rng.choice([10, 100, 300])
        kw.update(req_drop_permille=d, reply_drop_permille=rng.choice([0, d]))
    if rng.random() < 0.3:
        kw["reply_delay_permille"] = 667
    if rng.random() < 0.8:
        period = rng.choice([1, 2, 3, 5, 9])
        kw.update(start_period=period, start_first_tick=rng.choice([0, 7, 20]), start_max=kw["log_cap"] - rng.choice([1, 5, 20]),
                  start_mode=rng.choice([_abi.START_ONE, _abi.START_ALL]))
    if rng.random() < 0.6:
        kw.update(part_period=rng.choice([23, 60, 100]), part_jitter=rng.choice([0, 7, 20]))
    if rng.random() < 0.5:
        kw.update(churn_period=rng.choice([35, 70]), churn_disconnect_permille=200, churn_restart_permille=500, churn_crash_permille=200)
    if rng.random() < 0.3:
        kw["account_bytes"] = 1
    return kw


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RAFT_FUZZ_SEEDS", "48"))))
def test_random_schedule(seed):
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    rng = random.Random(1000 + seed)
    kw = _random_config(rng)
    o, e = OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))
    t = 0
    while t < 1500:
        n = rng.choice([1, 1, 2, 3, 5, 8, 13, 50, 200, 600])
        o.step(n)
        e.step(n)
        t += n
        d = diff_states(o, e)
        assert not d, "config %r after tick %d: %s" % (kw, t, "; ".join(d[:5]))
        if rng.random() < 0.2:                      # host verbs between launches
            g, p = rng.randrange(kw["n_groups"]), rng.randrange(kw["n_peers"])
            verb = rng.choice(["start", "disconnect", "connect", "crash", "restart"])
            for x in (o, e):
                if verb == "start":
                    x.start(g, p, 12345 + t)
                elif verb == "disconnect":
                    x.set_connected(g, p, False)
                elif verb == "connect":
                    x.set_connected(g, p, True)
                elif verb == "crash":
                    x.crash(g, p)
                else:
                    x.crash(g, p)
                    x.restart(g, p)
    assert np.array_equal(o.state_hash(), e.state_hash())
    assert o.safety_check() == e.safety_check()
