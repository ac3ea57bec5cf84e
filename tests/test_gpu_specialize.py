"""rg_specialize: the tick kernel compiled at run time for one engine's configuration gives the generic kernel's -- and
the CPU oracle's -- results, word for word, also after what was compiled in has changed (-m gpu)."""
import numpy as np
import pytest

from raft_amd import _abi
from tests.parity import diff_states

pytestmark = pytest.mark.gpu

LOSSY = dict(req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=90, part_period=60, part_jitter=20)
CASES = {
    "lean_5": dict(n_groups=40, n_peers=5, log_cap=128, seed=3, **LOSSY),
    "lean_3": dict(n_groups=50, n_peers=3, log_cap=96, seed=4, **LOSSY),
    "churn_7": dict(n_groups=24, n_peers=7, log_cap=256, seed=5, start_mode=_abi.START_ALL, churn_period=35,
                    churn_disconnect_permille=200, churn_restart_permille=500, churn_crash_permille=200, **LOSSY),
    "bytes_and_ring": dict(n_groups=30, n_peers=5, log_cap=128, seed=6, account_bytes=1, apply_ring_cap=64, **LOSSY),
    "lossless_reorder": dict(n_groups=30, n_peers=4, log_cap=128, seed=7, reply_delay_permille=667, start_period=3, start_max=100),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_specialized_kernel_equals_oracle(name):
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    kw = CASES[name]
    o, e = OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))
    e.specialize()
    for n in (1, 7, 300, 900):
        o.step(n)
        e.step(n)
        d = diff_states(o, e)
        assert not d, "%s after tick %d: %s" % (name, e.now(), "; ".join(d[:5]))
    assert o.counters() == e.counters()
    if kw.get("account_bytes"):
        assert o.rpc_bytes() == e.rpc_bytes()


def test_rebuilt_when_the_configuration_changes():
    """Rates changed by rg_set_unreliable, then a schedule (another instantiation): each next launch runs a kernel
    compiled for what holds then."""
    from oracle.binding import OracleEngine
    from raft_amd import Engine
    kw = CASES["lean_5"]
    o, e = OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))
    e.specialize()
    for x in (o, e):
        x.step(200)
        x.set_unreliable(300, 0, 667)
        x.step(200)
        x.set_schedule([(450, _abi.OP_PICK_LEADER, 0, 0), (450, _abi.OP_DISCONNECT, _abi.SEL_REG(0, 0), 0), (700, _abi.OP_CONNECT, _abi.SEL_ALL, 0)])
        x.step(500)
    o.step(50)
    e.step_noalloc(50)          # uses the kernel that is built and fits; compiles nothing itself
    assert not diff_states(o, e)
    assert o.counters() == e.counters()


def test_unknown_compiler_options_are_refused_not_fatal(monkeypatch):
    """RAFTGPU_SPEC_DEFS is a tuning aid that reaches a compiler whose option parser ends the process on an option it
    does not know.  The library passes on only the options it lists: anything else is RG_ERR_ARG, the process lives
    and the engine keeps running on the generic kernel."""
    from oracle.binding import OracleEngine
    from raft_amd import Engine, RaftGpuError
    kw = CASES["lean_5"]
    for bad in ("-mllvm -no-such-backend-switch=1", "-mllvm", "--version", "-DFOO=1", "-DRG_RING=4;x", "-mllvm -amdgpu-sched-strategy=bogus",
                "-O3 -include /etc/passwd"):
        monkeypatch.setenv("RAFTGPU_SPEC_DEFS", bad)
        e = Engine(_abi.make_config(**kw))
        with pytest.raises(RaftGpuError) as ei:
            e.specialize()
        assert ei.value.status == _abi.ERR_ARG, bad
        e.step(50)                               # the generic kernel serves
        assert e.now() == 50
        e.close()
    # options on the list go through, and the kernel they build is still the same function of the state
    monkeypatch.setenv("RAFTGPU_SPEC_DEFS", "-DRG_VERIFY_BATCH=32 -fno-unroll-loops -mllvm -amdgpu-sched-strategy=max-ilp")
    o, e = OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))
    e.specialize()
    o.step(600)
    e.step(600)
    assert not diff_states(o, e)
    # opting every engine in through the environment with a bad option: said once on stderr, the generic kernel runs
    monkeypatch.setenv("RAFTGPU_SPEC_DEFS", "-mllvm -no-such-backend-switch=1")
    monkeypatch.setenv("RAFTGPU_SPECIALIZE", "1")
    o, e = OracleEngine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))
    o.step(300)
    e.step(300)
    assert not diff_states(o, e)


def test_bench_workload_specialized_equals_generic():
    """Full size: 65,536 five-peer groups, the benchmark's 20,000 ticks, every group's digest."""
    from raft_amd import Engine
    kw = dict(n_groups=65536, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100,
              start_period=5, start_max=1000, part_period=100, part_jitter=20)
    a, b = Engine(_abi.make_config(**kw)), Engine(_abi.make_config(**kw))
    b.specialize()
    a.step(20000)
    b.step(20000)
    assert np.array_equal(a.state_hash(), b.state_hash())
    assert a.counters() == b.counters()
    assert a.safety_flags() == b.safety_flags()
    (ma, na), (mb, nb) = a.step_time_ms(), b.step_time_ms()
    print("generic %.2f ms, specialised %.2f ms" % (ma / na, mb / nb))
