#!/usr/bin/env python3
"""Compares the output of the Go lockstep driver (bindings/go/lockstep/lockstep_test.go, run inside the
reference's src/raft/ on a host with Go) with tests/golden/lockstep_digests.json, which the CPU restatement wrote.
Equal files pin the restatement -- and through it the HIP engine -- to the reference's own RequestVote /
AppendEntries / Start / persist code on these schedules.

usage: python tests/golden/diff_go_lockstep.py <go_lockstep.json> [golden.json]      exit status 0 = equal
"""
import json
import os
import sys

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lockstep_digests.json")


def differences(go, golden):
    """List of (case, tick, key, go value, golden value), in order; counters are compared one by one so that the first
    diverging quantity shows, the state digest last."""
    out = []
    for case in sorted(golden):
        if case not in go:
            out.append((case, None, "case missing from the Go output", None, None))
            continue
        if len(go[case]) != len(golden[case]):
            out.append((case, None, "number of stops", len(go[case]), len(golden[case])))
            continue
        for a, b in zip(go[case], golden[case]):
            tick = b["tick"]
            if a.get("tick") != tick:
                out.append((case, tick, "tick", a.get("tick"), tick))
                continue
            for k in sorted(b["counters"]):
                if a.get("counters", {}).get(k) != b["counters"][k]:
                    out.append((case, tick, "counters." + k, a.get("counters", {}).get(k), b["counters"][k]))
            for k in sorted(b):
                if k in ("counters", "tick", "state_sha256"):
                    continue
                if a.get(k) != b[k]:
                    out.append((case, tick, k, a.get(k), b[k]))
            if a.get("state_sha256") != b["state_sha256"]:
                out.append((case, tick, "state_sha256", a.get("state_sha256"), b["state_sha256"]))
    for case in sorted(set(go) - set(golden)):
        out.append((case, None, "case not in the golden file", None, None))
    return out


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    go = json.load(open(argv[1]))
    golden = json.load(open(argv[2] if len(argv) > 2 else GOLDEN))
    d = differences(go, golden)
    for case, tick, key, x, y in d[:40]:
        print("%s @%s: %s: go %r, restatement %r" % (case, tick, key, x, y))
    print("%d difference(s) in %d case(s)" % (len(d), len({c for c, *_ in d})) if d else "equal: %d cases" % len(golden))
    return 1 if d else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
