"""HIP engine vs CPU oracle, bit for bit, per group and per tick (-m gpu)."""
import numpy as np
import pytest

from raft_amd import _abi
from tests.parity import diff_states, run_lockstep

pytestmark = pytest.mark.gpu


def _pair(**kw):
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    return OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))


def test_boot_lossless_every_tick():
    o, e = _pair(n_groups=5, n_peers=3, log_cap=64, seed=1, start_period=5, start_first_tick=20, start_max=20)
    run_lockstep(o, e, [1] * 200)
    assert o.counters()["entries_committed"] == 5 * 20


@pytest.mark.parametrize("peers", [3, 5, 7, 4, 2, 8, 6])
def test_lossless_workload(peers):
    o, e = _pair(n_groups=37, n_peers=peers, log_cap=128, seed=3, start_period=5, start_first_tick=0, start_max=100)
    run_lockstep(o, e, [1] * 60 + [7] * 40 + [100] * 5)
    assert e.safety_check() == (0, 0)


@pytest.mark.parametrize("peers,seed", [(5, 2), (3, 9), (7, 4)])
def test_unreliable_rolling_partitions(peers, seed):
    o, e = _pair(n_groups=64, n_peers=peers, log_cap=256, seed=seed, req_drop_permille=100, reply_drop_permille=100,
                 start_period=5, start_max=200, part_period=100, part_jitter=20, account_bytes=1)
    run_lockstep(o, e, [1] * 300 + [13] * 100 + [500] * 4)
    assert o.safety_check() == e.safety_check()


@pytest.mark.parametrize("peers,seed,mode", [(7, 5, _abi.START_ALL), (5, 6, _abi.START_ALL), (3, 7, _abi.START_ONE)])
def test_churn_crash_restart(peers, seed, mode):
    o, e = _pair(n_groups=48, n_peers=peers, log_cap=512, seed=seed, req_drop_permille=50, reply_drop_permille=50,
                 start_period=3, start_max=400, start_mode=mode, churn_period=70, churn_disconnect_permille=200,
                 churn_restart_permille=500, churn_crash_permille=200)
    run_lockstep(o, e, [1] * 400 + [10] * 100 + [1000] * 3)
    assert o.safety_check() == e.safety_check()


def test_long_reordering_and_heavy_loss():
    o, e = _pair(n_groups=32, n_peers=5, log_cap=256, seed=11, req_drop_permille=300, reply_drop_permille=300,
                 reply_delay_permille=667, start_period=2, start_max=250, start_mode=_abi.START_ALL,
                 part_period=37, part_jitter=11, account_bytes=1)
    run_lockstep(o, e, [1] * 500 + [250] * 6)
    assert o.rpc_bytes(100) == e.rpc_bytes(100) > 0
