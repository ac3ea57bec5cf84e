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


def _free_bytes():
    """Free device memory as the driver reports it: hipMemGetInfo of the HIP runtime the library itself is linked
    against (already loaded with it; no second runtime comes into the process)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


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
    # and onwards from there, in the other order, long enough to go through the relay again
    more = 2400 if ticks >= 2000 else 500
    a.step_noalloc(more)
    b.step(more)
    assert np.array_equal(a.state_hash(), b.state_hash())
    assert a.counters() == b.counters()
    a.close()
    b.close()


def test_noalloc_allocates_nothing():
    """What the entry point exists for: after a first call, further calls -- relay launches through the fixed array
    and plain ones -- leave the device's free memory exactly where it was (rg_step, for contrast, creates events)."""
    from raft_amd import Engine
    big = Engine(_abi.make_config(n_groups=65536, **KW))
    small = Engine(_abi.make_config(n_groups=3000, **KW))
    big.step_noalloc(2000)
    small.step_noalloc(300)
    big.sync()
    small.sync()
    free0 = _free_bytes()
    for _ in range(3):
        big.step_noalloc(2500)       # relay launch
        small.step_noalloc(400)      # plain launch
        big.step_noalloc(7)          # plain launch of a large engine
        big.step_noalloc(2)          # due launch: only the groups with an event inside it
    big.sync()
    small.sync()
    assert _free_bytes() == free0
    big.close()
    small.close()


def test_two_engines_share_the_array_without_racing():
    """Two large engines on one GPU, each on its own stream, stepping through the entry point back to back with no
    synchronisation in between: the second launch waits on the GPU for the first (a per-device event), so neither
    sees the other's queue or staging words.  Both must end where rg_step takes an engine of their configuration."""
    from raft_amd import Engine
    cfgs = [dict(KW, n_groups=65536, seed=21), dict(KW, n_groups=60000, seed=22, n_peers=3, log_cap=128, start_max=100)]
    ref, got = [], []
    for kw in cfgs:
        r = Engine(_abi.make_config(**kw))
        r.step(6000)
        ref.append((r.state_hash(), r.counters()))
        r.close()
    engs = [Engine(_abi.make_config(**kw)) for kw in cfgs]
    for _ in range(3):                # 3 x 2,000 ticks each, interleaved, nothing waits on the host
        for e in engs:
            e.step_noalloc(2000)
    for e in engs:
        got.append((e.state_hash(), e.counters()))
        e.close()
    for (hr, cr), (hg, cg) in zip(ref, got):
        assert np.array_equal(hr, hg) and cr == cg


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
    e.step_noalloc(500)     # too short for a relay launch: no scratch needed, not refused
    assert e.now() == 500
    e.step(10)              # the engine is otherwise fine
    assert e.now() == 510
    e.close()
