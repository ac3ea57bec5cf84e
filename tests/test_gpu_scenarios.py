"""The reference's Lab 2A/2B/2C tests and the hand-derived known answers,
through the C ABI on the HIP engine (-m gpu)."""
import pytest

from raft_amd import _abi
from raft_amd.harness import TestFailure
from tests import kat, scenarios

pytestmark = pytest.mark.gpu


@pytest.fixture
def mk():
    from raft_amd import Engine
    return lambda **kw: Engine(_abi.make_config(**kw))


@pytest.mark.parametrize("case", sorted(kat.CASES))
def test_known_answer(mk, case):
    kat.CASES[case](mk)


@pytest.mark.parametrize("name", [n for n in scenarios.ALL if n not in ("ReElection2A", "Figure8Unreliable2C")])
def test_scenario(mk, name):
    scenarios.ALL[name](mk)


@pytest.mark.parametrize("name", sorted(scenarios.EXTRA))
def test_binding_model_matches_the_checker(name):
    """Several configs on one engine and the labrpc-shaped ends: same asserts, and the same end state as the checker."""
    import numpy as np
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    from tests.parity import diff_states
    o = scenarios.EXTRA[name](lambda **kw: OracleEngine(_abi.make_config(**kw)))
    e = scenarios.EXTRA[name](lambda **kw: Engine(_abi.make_config(**kw)))
    assert o.now() == e.now()
    assert not diff_states(o, e)
    assert np.array_equal(o.state_hash(), e.state_hash())


def test_reelection_split_vote_never_resolves(mk):
    with pytest.raises(TestFailure, match="expected one leader, got none"):
        scenarios.reelection_2a(mk)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_figure8_unreliable(mk, seed):
    try:
        scenarios.figure8_unreliable_2c(mk, seed=seed)
    except TestFailure as ex:
        assert "commit index=" in str(ex) or "committed values do not match" in str(ex)


def test_scenarios_end_in_the_same_state_as_the_checker():
    """Host-driven runs (Start, connect, crash, restart between steps) leave the
    HIP engine and the CPU checker in the same state, word for word."""
    import numpy as np
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    from tests.parity import diff_states
    for name in ("Backup2B", "Persist22C", "Figure82C", "UnreliableChurn2C"):
        made = []

        def mko(**kw):
            made.append(OracleEngine(_abi.make_config(**kw)))
            return made[-1]

        def mke(**kw):
            made.append(Engine(_abi.make_config(**kw)))
            return made[-1]
        scenarios.ALL[name](mko)
        scenarios.ALL[name](mke)
        o, e = made
        assert o.now() == e.now(), name
        d = diff_states(o, e)
        assert not d, "%s: %s" % (name, d[:5])
        assert np.array_equal(o.state_hash(), e.state_hash()), name
