"""A replicated key/value service on top of engine groups, producing client histories for the
linearizability checker.

The reference's own service layer is an unimplemented stub (src/kvraft/server.go:41-47,
src/kvraft/client.go:39-70); this module supplies the minimum the lab text of those stubs asks for, so
that sampled groups can be judged the way src/kvraft/test_test.go:296-400 judges a real service: every
server applies the committed operations in log order to a map, duplicates of a retried operation are
dropped by (client, sequence number), a Clerk retries other servers until its operation is applied at the
index its Start returned.  Client workload and history records follow GenericTestLinearizability
(test_test.go:341-371).  Everything runs in lockstep ticks: `KvCluster.tick()` = clients act, one engine
step, servers drain their apply streams.

Works with any EngineBase (the HIP engine, or the CPU checker in tests).
"""
import random

from .lincheck import APPEND, GET, PUT, Operation


class KVServer:
    """State machine of one server (src/kvraft/server.go:33-48): applies drained ApplyMsgs in order."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.kv = {}
        self.last_seq = {}        # client -> highest sequence number applied
        self.last_index = 0

    def apply(self, index, op):
        """op = (client, seq, kind, key, value).  Returns the operation's output at this point of the log."""
        client, seq, kind, key, value = op
        if kind == GET:
            return self.kv.get(key, "")
        if seq > self.last_seq.get(client, -1):
            self.last_seq[client] = seq
            self.kv[key] = value if kind == PUT else self.kv.get(key, "") + value
        return ""


class Clerk:
    """One client (src/kvraft/client.go:8-70), non-blocking: `pending` is the operation in flight."""

    def __init__(self, cid, n_servers, rng):
        self.cid, self.n_servers, self.rng = cid, n_servers, rng
        self.leader = rng.randrange(n_servers)
        self.seq = 0
        self.count = 0            # puts and appends issued, the `j` of test_test.go:342
        self.pending = None       # dict(op, handle, call, server, index, since)


class KvGroup:
    def __init__(self, engine, g, n_clients, seed, retry_ticks=40, op_permille=300):
        self.e, self.g, self.P = engine, g, engine.P
        self.rng = random.Random(seed * 1000003 + g)
        self.servers = [KVServer() for _ in range(self.P)]
        self.clerks = [Clerk(c, self.P, self.rng) for c in range(n_clients)]
        self.ops = {}             # command handle -> (client, seq, kind, key, value): the engine carries u64 handles
        self.history = []
        self.retry_ticks, self.op_permille = retry_ticks, op_permille
        self.unfinished = 0

    # -- clients, before the step of tick t
    def clients_act(self, t, issue_new=True):
        n = len(self.clerks)
        for ck in self.clerks:
            if ck.pending is None:
                if not issue_new or self.rng.randrange(1000) >= self.op_permille:
                    continue
                key = str(self.rng.randrange(n))                              # test_test.go:347
                nv = "x %d %d y" % (ck.cid, ck.count)                         # test_test.go:348
                if self.rng.randrange(1000) < 500:
                    kind, value = APPEND, nv
                elif self.rng.randrange(1000) < 100:
                    kind, value = PUT, nv
                else:
                    kind, value = GET, ""
                if kind != GET:
                    ck.count += 1
                handle = ((ck.cid + 1) << 32) | ck.seq
                self.ops[handle] = (ck.cid, ck.seq, kind, key, value)
                ck.pending = dict(kind=kind, key=key, value=value, handle=handle, call=2 * t, server=None, index=0, since=t)
                ck.seq += 1
            p = ck.pending
            if p["server"] is not None and t - p["since"] > self.retry_ticks:   # no answer: try the next server
                p["server"] = None
                ck.leader = (ck.leader + 1) % self.P
            if p["server"] is None:
                index, _term, is_leader = self.e.start(self.g, ck.leader, p["handle"])
                if is_leader:
                    p["server"], p["index"], p["since"] = ck.leader, index, t
                else:
                    ck.leader = (ck.leader + 1) % self.P

    # -- servers, after the step of tick t
    def servers_apply(self, t):
        waiting = {}
        for ck in self.clerks:
            if ck.pending and ck.pending["server"] is not None:
                waiting.setdefault(ck.pending["server"], []).append(ck)
        for s, cks in waiting.items():
            srv = self.servers[s]
            for index, handle in self.e.drain_applied(self.g, s):
                if index <= srv.last_index:          # the peer restarted and replays its log (raft.go:672-688 from lastApplied 0)
                    srv.reset()
                srv.last_index = index
                op = self.ops.get(handle)
                out = srv.apply(index, op) if op is not None else ""
                for ck in cks:
                    p = ck.pending
                    if p is None or p["server"] != s or p["index"] != index:
                        continue
                    if handle == p["handle"]:
                        self.history.append(Operation(ck.cid, p["kind"], p["key"], p["value"], out, p["call"], 2 * t + 1))
                        ck.pending = None
                    else:                             # another leader's entry took the index: submit again
                        p["server"] = None
                        ck.leader = (ck.leader + 1) % self.P

    def finish(self, t_end):
        """The history.  A put or append still in flight may or may not have taken effect: it is recorded as
        returning after everything else, so the checker is free to place it last."""
        out = list(self.history)
        self.unfinished = 0
        for ck in self.clerks:
            p = ck.pending
            if p is None:
                continue
            self.unfinished += 1
            if p["kind"] != GET:
                out.append(Operation(ck.cid, p["kind"], p["key"], p["value"], "", p["call"], 2 * t_end + 10 + ck.cid))
        return out


class KvCluster:
    """KvGroups on sampled groups of one engine, advanced together."""

    def __init__(self, engine, groups, n_clients=5, seed=1, **kw):
        self.e = engine
        self.groups = [KvGroup(engine, g, n_clients, seed, **kw) for g in groups]
        self.t = 0

    def tick(self, issue_new=True):
        for kg in self.groups:
            kg.clients_act(self.t, issue_new)
        self.e.step(1)
        for kg in self.groups:
            kg.servers_apply(self.t)
        self.t += 1

    def run(self, ticks, drain_ticks=300):
        for _ in range(ticks):
            self.tick()
        for _ in range(drain_ticks):                 # let the operations in flight finish; no new ones
            if not any(ck.pending for kg in self.groups for ck in kg.clerks):
                break
            self.tick(issue_new=False)
        return {kg.g: kg.finish(self.t) for kg in self.groups}
