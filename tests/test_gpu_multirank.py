"""The sharded path with HIP engines in two processes (-m gpu): each rank owns a block of global group ids on
GPU 0, the summary is reduced by the product's all_reduce_summary (over gloo here: two ranks cannot share one
GPU under RCCL), and the result must equal one engine holding both blocks.  Also rehearses bench.py's own
rank launcher end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raft_amd import _abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(n_peers=5, log_cap=128, seed=2, req_drop_permille=100, reply_drop_permille=100,
          start_period=5, start_max=100, part_period=100, part_jitter=20)
G, TICKS = 600, 1500


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raft_amd import Engine
    from raft_amd.distributed import all_reduce_words, shard_base
    e = Engine(_abi.make_config(n_groups=G, group_base=shard_base(rank, G), device=0, **KW))
    e.step(TICKS)
    e.safety_check()
    res = all_reduce_words(*e.allreduce_summary(), torch.device("cpu"))     # the engine's packed words (no communicator), summed over gloo
    torch.save((res, e.state_hash().tolist()), "%s.%d" % (out, rank))
    e.close()
    dist.destroy_process_group()


def test_two_hip_ranks_equal_one_engine(tmp_path):
    from raft_amd import Engine
    out = str(tmp_path / "rank")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (res0, h0), (res1, h1) = torch.load(out + ".0"), torch.load(out + ".1")
    assert res0 == res1                                   # every rank holds the reduced summary
    from raft_amd.distributed import flags_of
    counters, summary = res0
    whole = Engine(_abi.make_config(n_groups=2 * G, **KW))
    whole.step(TICKS)
    whole.safety_check()
    assert counters == whole.counters()
    assert summary == whole.safety_summary()
    assert flags_of(summary) == whole.safety_flags()
    assert h0 + h1 == whole.state_hash().tolist()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it prints one line for the whole job."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "1", "--warmup", "0",
           "--groups", "3000", "--ticks", "1500", "--entries", "200", "--log-cap", "256", "--no-cpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["groups_total"] == 6000
    assert out["counters_per_step"]["group_ticks"] == 2 * 3000 * 1500
    assert "rehearsal" in out
    # the same two shards as one engine
    from raft_amd import Engine
    e = Engine(_abi.make_config(n_groups=6000, n_peers=5, log_cap=256, seed=2, req_drop_permille=100, reply_drop_permille=100,
                                start_period=5, start_max=200, part_period=100, part_jitter=20))
    e.step(1500)
    assert out["counters_per_step"] == e.counters()
    assert [out["safety_flags"], out["flagged_groups"]] == list(e.safety_flags())
    assert out["safety"]["flagged_groups_by_bit"] == {n: e.safety_summary()["groups_" + n] for n in _abi.FLAG_NAMES}
    assert "torch.distributed" in out["config"]["collective"]
