"""rg_step_noalloc (the entry point that allocates nothing: relay queue and staging area in a fixed device
array) against rg_step, through the C ABI: every group's state digest, the flag words and the counters must be
equal.  Two calls in a row on one GPU share the array; the second engine is smaller, so stale words of the
first call lie behind and between the ones it uses."""
import numpy as np
import pytest

from raft_amd import _abi

pytestmark = pytest.mark.gpu

KW = dict(n_peers=5, log_cap=256, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5,
          start_max=200, part_period=100, part_jitter=20)


def _pair(groups, ticks, **over):
    from raft_amd import Engine
    kw = dict(KW, **over)
    a, b = Engine(_abi.make_config(n_groups=groups, **kw)), Engine(_abi.make_config(n_groups=groups, **kw))
    a.step(ticks)
    b.step_noalloc(ticks)
    assert b.now() == ticks
    assert np.array_equal(a.state_hash(), b.state_hash())
    assert np.array_equal(a.read("gflags"), b.read("gflags"))
    assert a.counters() == b.counters()
    # and onwards from there, in the other order
    a.step_noalloc(500)
    b.step(500)
    assert np.array_equal(a.state_hash(), b.state_hash())
    assert a.counters() == b.counters()
    a.close()
    b.close()


def test_noalloc_equals_step_twice_second_smaller():
    _pair(65536, 4000)      # relay launch: 1,366 sets x 4 chunks through the fixed array
    _pair(50000, 2500)      # fewer sets, other chunk length: the array still holds the first call's words
    _pair(3000, 700)        # fits the chip at once: plain launch, no scratch used
    _pair(60000, 3000, n_peers=3, log_cap=128, start_max=100)   # three-peer sets have another staging stride


def test_noalloc_refuses_engines_beyond_the_scratch():
    from raft_amd import Engine
    from raft_amd.engine import RaftGpuError
    e = Engine(_abi.make_config(n_groups=230000, n_peers=5, log_cap=16, seed=1))
    with pytest.raises(RaftGpuError) as ei:
        e.step_noalloc(3000)
    assert ei.value.status == _abi.ERR_SCRATCH
    assert e.now() == 0
    e.step(10)              # the engine is otherwise fine
    assert e.now() == 10
    e.close()
