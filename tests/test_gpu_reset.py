"""rg_reset (-m gpu): an engine that has run, been reset and run again holds what a fresh engine holds after the second
run -- although the reset leaves the old run's entries in the log rows beyond the (new) lengths and the old tester
records in place."""
import numpy as np
import pytest

from raft_amd import _abi
from tests.parity import diff_states
from tests.test_gpu_due import CHURN, LOSSY, THREE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw,groups,first,second", [(LOSSY, 900, 1200, 700), (CHURN, 300, 900, 1100), (THREE, 1500, 600, 450)])
def test_second_run_after_reset_equals_a_fresh_run(kw, groups, first, second):
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    cfg = dict(n_groups=groups, **kw)
    e = Engine(_abi.make_config(**cfg))
    e.step(first)                                  # long logs, records, flags, counters: all to be forgotten
    assert e.counters()["entries_committed"] > 0
    for round_ in range(2):                        # twice: the second reset meets what the first second run left
        e.reset()
        assert e.now() == 0 and e.counters()["entries_committed"] == 0
        o = OracleEngine(_abi.make_config(**cfg))
        for n in (1, 7, second // 2, 3, second - second // 2 - 11):
            o.step_mt(n, 8)
            e.step(n)
        d = diff_states(o, e)
        assert not d, "run %d after reset: %s" % (round_, "; ".join(d[:6]))
        assert np.array_equal(o.state_hash(), e.state_hash())
        assert o.safety_check() == e.safety_check()
        if kw.get("apply_ring_cap"):
            from raft_amd.engine import RaftGpuError
            for g in (0, groups // 2, groups - 1):
                for p in range(kw["n_peers"]):
                    res = []
                    for x in (o, e):
                        try:
                            res.append(("ok", x.drain_applied(g, p)))
                        except RaftGpuError as err:     # more messages than the ring holds: the newest ones, and the status
                            res.append((err.status, err.partial))
                    assert res[0] == res[1]
