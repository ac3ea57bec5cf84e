"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run): every rank gets the
rendezvous environment, rank 0's line is passed on, a failing rank fails the run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RANK_SCRIPT = """
import json, os, sys
env = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
open(sys.argv[1] + "." + env["RANK"], "w").write(json.dumps(env))
if env["RANK"] == "0":
    print(json.dumps({"rank0": True, "argv": sys.argv[2:]}))
sys.exit(int(sys.argv[2]) if env["RANK"] == sys.argv[3] else 0)
"""


def test_launcher_sets_rendezvous_env_and_relays_rank0(tmp_path, capfd):
    import bench
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    out = str(tmp_path / "env")
    rc = bench.launch_ranks(3, script=str(script), argv=[out, "0", "-1"], check_devices=False)
    assert rc == 0
    line = capfd.readouterr().out.strip()
    assert json.loads(line) == {"rank0": True, "argv": ["0", "-1"]}
    envs = [json.loads(open(out + ".%d" % r).read()) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1


def test_launcher_reports_a_failing_rank(tmp_path, capfd):
    import bench
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    rc = bench.launch_ranks(2, script=str(script), argv=[str(tmp_path / "env"), "7", "1"], check_devices=False)
    assert rc == 7
