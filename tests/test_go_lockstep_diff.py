"""The script that holds the Go lockstep driver's output against the golden file, on hand-made outputs (the driver
itself needs a Go toolchain and the reference's sources: it does not run here)."""
import copy
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("diff_go_lockstep", os.path.join(HERE, "golden", "diff_go_lockstep.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)


def test_equal_and_perturbed_outputs(tmp_path):
    golden = json.load(open(mod.GOLDEN))
    assert mod.differences(copy.deepcopy(golden), golden) == []
    go = copy.deepcopy(golden)
    case = sorted(go)[1]
    go[case][0]["counters"]["persists"] += 1
    go[case][0]["state_sha256"] = "0" * 64
    del go[sorted(go)[0]]
    d = mod.differences(go, golden)
    assert d[0][2] == "case missing from the Go output"
    assert (d[1][0], d[1][2]) == (case, "counters.persists") and d[-1][2] == "state_sha256" or d[2][2] == "state_sha256"
    out = tmp_path / "go.json"
    out.write_text(json.dumps(go))
    assert mod.main(["diff", str(out)]) == 1
    out.write_text(json.dumps(golden))
    assert mod.main(["diff", str(out)]) == 0


def test_cases_file_matches_the_golden_file():
    """lockstep_cases.json (what the Go driver reads) lists exactly the golden cases and stops."""
    golden = json.load(open(mod.GOLDEN))
    cases = json.load(open(os.path.join(HERE, "golden", "lockstep_cases.json")))
    assert sorted(c["name"] for c in cases) == sorted(golden)
    for c in cases:
        assert c["stops"] == [s["tick"] for s in golden[c["name"]]]
        assert c["config"]["n_peers"] >= 3
