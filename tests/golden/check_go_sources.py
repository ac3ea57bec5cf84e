#!/usr/bin/env python3
"""Keeps the Go side honest where no Go compiler exists.

The cgo binding (bindings/go/raft/*.go) and the lockstep driver (bindings/go/lockstep/lockstep_test.go) are
written against include/raftgpu.h and oracle/raft_oracle.c but cannot be compiled in this image.  This script
checks, from the texts alone:

  1. every `C.rg_*(...)` call of the binding names a function the header declares, passes as many arguments as
     the prototype has, and passes a pointer exactly where the prototype has one (as far as the argument's
     spelling tells: `&x`, `nil`, `unsafe.Pointer(..)` are pointers; `C.int32_t(..)`, literals are not);
  2. every `C.RG_*` constant of the binding is defined in the header;
  3. the lockstep driver's constants -- timing literals, flag bits, stream numbers, the counter list and its
     order -- equal the header's and the oracle's.

Run:  python tests/golden/check_go_sources.py     (exit status 0 = consistent; tests/test_go_sources.py runs it)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HEADER = os.path.join(ROOT, "include", "raftgpu.h")
ORACLE = os.path.join(ROOT, "oracle", "raft_oracle.c")
GO_DIR = os.path.join(ROOT, "bindings", "go", "raft")
LOCKSTEP = os.path.join(ROOT, "bindings", "go", "lockstep", "lockstep_test.go")


def _strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _split_args(s):
    """Top-level comma split of an argument list."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def header_prototypes(text=None):
    """name -> [is_pointer per parameter] for every rg_* function the header declares."""
    text = _strip_c_comments(text if text is not None else open(HEADER).read())
    protos = {}
    for m in re.finditer(r"\b(?:int|const\s+char\s*\*)\s*(rg_\w+)\s*\(([^;{]*)\)\s*;", text):
        params = m.group(2).strip()
        args = [] if params in ("", "void") else _split_args(params)
        protos[m.group(1)] = ["*" in a for a in args]
    return protos


def header_constants(text=None):
    """#define and enum names of the header -> integer value where it is a plain one."""
    raw = text if text is not None else open(HEADER).read()
    text = _strip_c_comments(raw)
    consts = {}
    for m in re.finditer(r"#define\s+(RG_\w+)(?:\(.*?\))?[ \t]*([^\n]*)", text):
        v = m.group(2).strip().rstrip("u")
        try:
            consts[m.group(1)] = int(v, 0)
        except ValueError:
            consts[m.group(1)] = None
    for m in re.finditer(r"enum\s*\{(.*?)\}", text, flags=re.S):
        nxt = 0
        for item in _split_args(m.group(1)):
            if not item:
                continue
            name, _, val = item.partition("=")
            name = name.strip()
            if val.strip():
                v = val.strip()
                nxt = consts.get(v, None) if not re.match(r"^-?(0x)?[0-9a-fA-F]+$", v) else int(v, 0)
            consts[name] = nxt
            if nxt is not None:
                nxt += 1
    return consts


def _classify(arg):
    """'ptr', 'val' or None (cannot tell from the spelling)."""
    a = arg.strip()
    if a.startswith("&") or a == "nil" or a.startswith("unsafe.Pointer(") or a.startswith("(*C."):
        return "ptr"
    if re.match(r"^C\.(u?int\d+_t|int|double)\(", a) or re.match(r"^-?\d+$", a):
        return "val"
    if re.match(r"^C\.RG_\w+$", a):
        return "val"
    return None


def go_calls(path):
    """(line, name, [args]) for every C.rg_* call in a Go file."""
    text = open(path).read()
    calls = []
    for m in re.finditer(r"\bC\.(rg_\w+)\(", text):
        i, depth = m.end(), 1
        while depth and i < len(text):
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        calls.append((text.count("\n", 0, m.start()) + 1, m.group(1), _split_args(text[m.end():i - 1])))
    return calls


# Go identifiers of the binding whose kind the spelling does not show
KNOWN = {"ptr": {"en.e", "e", "grp.en.e", "rf.grp.en.e", "p", "engineFor(n).e"},
         "val": {"g", "grp.g", "rf.grp.g", "me", "v", "drop", "delay", "size", "n"}}


def check_binding(errors):
    protos, consts = header_prototypes(), header_constants()
    used = set()
    for fn in sorted(os.listdir(GO_DIR)):
        if not fn.endswith(".go"):
            continue
        path = os.path.join(GO_DIR, fn)
        for line, name, args in go_calls(path):
            where = "%s:%d" % (os.path.relpath(path, ROOT), line)
            if name not in protos:
                errors.append("%s: C.%s is not declared in include/raftgpu.h" % (where, name))
                continue
            used.add(name)
            want = protos[name]
            if len(args) != len(want):
                errors.append("%s: C.%s takes %d arguments, called with %d" % (where, name, len(want), len(args)))
                continue
            for k, (a, is_ptr) in enumerate(zip(args, want)):
                kind = _classify(a) or ("ptr" if a in KNOWN["ptr"] else "val" if a in KNOWN["val"] else None)
                if kind == "ptr" and not is_ptr:
                    errors.append("%s: C.%s argument %d (%s) is a pointer, the prototype has a value" % (where, name, k + 1, a))
                if kind == "val" and is_ptr:
                    errors.append("%s: C.%s argument %d (%s) is a value, the prototype has a pointer" % (where, name, k + 1, a))
        for m in re.finditer(r"\bC\.(RG_\w+)", open(path).read()):
            if m.group(1) not in consts:
                errors.append("%s: C.%s is not defined in include/raftgpu.h" % (os.path.relpath(path, ROOT), m.group(1)))
    return used


def check_lockstep(errors):
    go = open(LOCKSTEP).read()
    consts = header_constants()
    oracle = open(ORACLE).read()

    def go_const(name):
        m = re.search(r"\b%s\s*=\s*(0x[0-9a-fA-F]+|\d+)" % re.escape(name), go)
        return int(m.group(1), 0) if m else None

    pairs = {"lsTickMs": "RG_TICK_MS", "lsElectionPeriod": "RG_ELECTION_PERIOD_MS", "lsLeaderExtra": "RG_ELECTION_LEADER_EXTRA_MS",
             "lsMonitorPoll": "RG_MONITOR_POLL_MS", "lsHbIdle": "RG_HB_IDLE_MS", "lsHbBacklog": "RG_HB_BACKLOG_MS",
             "lsCandidateWindow": "RG_CANDIDATE_WINDOW_MS", "lsTimeoutBase": "RG_TIMEOUT_BASE_MS", "lsTimeoutSpan": "RG_TIMEOUT_SPAN_MS",
             "lsRPCWait": "RG_RPC_WAIT_MS", "lsDelaySpan": "RG_NET_DELAY_SPAN_MS",
             "flagElectionSafety": "RG_FLAG_ELECTION_SAFETY", "flagLogMatching": "RG_FLAG_LOG_MATCHING",
             "flagLeaderCompleteness": "RG_FLAG_LEADER_COMPLETENESS", "flagStateMachine": "RG_FLAG_STATE_MACHINE",
             "flagApplyOrder": "RG_FLAG_APPLY_ORDER", "flagRefPanic": "RG_FLAG_REF_PANIC", "flagLogOverflow": "RG_FLAG_LOG_OVERFLOW"}
    for g, h in pairs.items():
        if go_const(g) is None:
            errors.append("lockstep_test.go: constant %s is missing" % g)
        elif go_const(g) != consts.get(h):
            errors.append("lockstep_test.go: %s = %r, include/raftgpu.h has %s = %r" % (g, go_const(g), h, consts.get(h)))
    # stream numbers against the oracle's
    for g, o in {"streamElection": "STREAM_ELECTION", "streamNet": "STREAM_NET", "streamCmd": "STREAM_CMD", "streamPart": "STREAM_PART",
                 "streamChurn": "STREAM_CHURN", "streamSched": "STREAM_SCHED"}.items():
        m = re.search(r"#define\s+%s\s+(\d+)ull" % o, oracle)
        if not m:
            errors.append("oracle/raft_oracle.c: %s is missing" % o)
        elif go_const(g) != int(m.group(1)):
            errors.append("lockstep_test.go: %s = %r, oracle/raft_oracle.c has %s = %s" % (g, go_const(g), o, m.group(1)))
    # the counter list and its order against the header's enum
    m = re.search(r"lsCounterNames\s*=\s*\[\]string\{(.*?)\}", go, flags=re.S)
    names = re.findall(r'"(\w+)"', m.group(1)) if m else []
    want = sorted((v, k) for k, v in consts.items() if k.startswith("RG_C_") and v is not None)
    want = [k[len("RG_C_"):].lower() for _, k in want]
    if names != want:
        errors.append("lockstep_test.go: lsCounterNames %r differ from the header's RG_C_* %r" % (names, want))
    # the one formula of the net model that is easy to get out of step: an unreachable Call's cost
    if "* 7000) >> 32" not in go or "7000ull) >> 32" not in oracle:
        errors.append("the cost of an unreachable Call (labrpc.go:296) is not spelled alike in lockstep_test.go and oracle/raft_oracle.c")


def main():
    errors = []
    used = check_binding(errors)
    check_lockstep(errors)
    for e in errors:
        print("MISMATCH", e)
    print("checked %d distinct C.rg_* functions of the binding against %d prototypes" % (len(used), len(header_prototypes())))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
