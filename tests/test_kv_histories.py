"""Client histories of a key/value service on sampled groups, judged by the linearizability checker
(src/kvraft/test_test.go:296-400 is the reference's form of this test; its service layer is a stub, so the
service here is raft_amd/kvservice.py).  CPU runs use the oracle engine, GPU runs the HIP engine, and both
must produce the same histories."""
import pytest

from raft_amd import _abi, lincheck as lc
from raft_amd.kvservice import KvCluster

NETS = {
    "reliable": {},
    "unreliable": dict(req_drop_permille=100, reply_drop_permille=100),                       # test_test.go:626-629
    "partitions": dict(req_drop_permille=100, reply_drop_permille=100, part_period=100, part_jitter=20),   # :650-653
}


SEEDS = {"partitions": 5}    # a seed under which one of the 8 groups loses a committed entry within 1500 ticks


def _histories(make_engine, net, groups=8, ticks=1500):
    e = make_engine(_abi.make_config(n_groups=groups, n_peers=5, log_cap=2048, apply_ring_cap=2048, seed=SEEDS.get(net, 3), **NETS[net]))
    kc = KvCluster(e, range(groups), n_clients=5, seed=1)
    hs = kc.run(ticks)
    flags = e.safety_check()
    return e, hs, e.read("gflags"), flags


def _judge(hs, gflags):
    res = {g: lc.check_operations_timeout(h, 10) for g, h in hs.items()}
    assert lc.UNKNOWN not in res.values()
    for g, r in res.items():
        # a group in which the reference's code kept its invariants serves a linearizable history
        if gflags[g] == 0:
            assert r == lc.OK, "group %d" % g
    return res


def _oracle(cfg):
    from oracle.binding import OracleEngine
    return OracleEngine(cfg)


def _engine(cfg):
    from raft_amd import Engine
    return Engine(cfg)


@pytest.mark.parametrize("net", ["reliable", "unreliable"])
def test_oracle_histories_are_linearizable(net):
    _, hs, gflags, flags = _histories(_oracle, net)
    assert flags == (0, 0)
    assert all(len(h) > 300 for h in hs.values())
    assert set(_judge(hs, gflags).values()) == {lc.OK}


def test_oracle_histories_under_partitions():
    """The reference loses committed entries under rolling partitions (its vote test and its persisted prefix
    both use commitIndex, raft.go:120,218-219): the group the engine flags serves a history the checker rejects,
    every other group a linearizable one."""
    _, hs, gflags, flags = _histories(_oracle, "partitions")
    res = _judge(hs, gflags)
    assert flags[1] >= 1
    assert any(r == lc.ILLEGAL for r in res.values())


def test_a_corrupted_history_is_rejected():
    _, hs, _, _ = _histories(_oracle, "reliable", groups=1, ticks=300)
    h = hs[0]
    assert lc.check_operations(h)
    k = next(i for i, o in enumerate(h) if o.op == lc.GET and o.output)
    h[k] = h[k]._replace(output=h[k].output + "!")
    assert not lc.check_operations(h)


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["reliable", "unreliable", "partitions"])
def test_gpu_histories_equal_the_oracle_s_and_are_judged_alike(net):
    _, ho, gfo, _ = _histories(_oracle, net, groups=6, ticks=1500)
    _, he, gfe, _ = _histories(_engine, net, groups=6, ticks=1500)
    assert ho == he
    assert gfo.tolist() == gfe.tolist()
    _judge(he, gfe)
