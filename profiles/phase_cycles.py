#!/usr/bin/env python3
"""Shader cycles per phase of a tick (diagnostic -DRG_STAMPS build).
usage: RAFTGPU_LIB=raft_amd/lib/libraftgpu_stamps.so python profiles/phase_cycles.py [idle|bench]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
G, T = int(os.environ.get("PG", 65536)), int(os.environ.get("PT", 4000))
mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
kw = dict(n_groups=G, n_peers=5, log_cap=1024, seed=2)
if mode == "bench":
    kw.update(req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=1000, part_period=100, part_jitter=20)
e = Engine(_abi.make_config(**kw))
e.step(T); e.sync()
buf = (C.c_uint64 * 56)()
e._lib.rg_debug_section_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert e._lib.rg_debug_section_counts(e._h, buf) == 0
def tick_waves(G, slots=4096, gw_max=12):
    """Waves of a five-peer launch under the engine's packing rule (fill_devcfg in raft_amd/csrc/raftgpu.hip)."""
    full = G // (slots * gw_max) * slots
    rest = G - full * gw_max
    spread = slots // 2 if full else slots
    gw = max(1, min(gw_max, -(-rest // spread)))
    return full + -(-rest // gw)


wt = tick_waves(G) * T
names = ["client Starts", "classify", "(rest of actors)", "apply", "leader check", "pend header+election", "RV section", "monitor eval", "AE section",
         "key + catch-up", "partitions, churn", "next event", "client Starts before a control event"]
cnt = {"pend header+election": 0, "RV section": 1, "monitor eval": 2, "AE section": 3, "apply": 4}
tot = sum(buf[40:53])
for i, n in enumerate(names):
    extra = ""
    if n in cnt and buf[_abi.N_COUNTERS + cnt[n]]:
        k = buf[_abi.N_COUNTERS + cnt[n]]
        extra = "  %.3f runs per wave-tick, %7.0f cycles per run" % (k / wt, buf[40 + i] / k)
    print("%-22s %8.1f cycles per wave-tick (%4.1f %%)%s" % (n, buf[40 + i] / wt, 100.0 * buf[40 + i] / tot, extra))
print("total                  %8.1f cycles per wave-tick" % (tot / wt))
extra = {7: "store fences", 8: "AE: matrix cache miss", 9: "AE: prevTerm from HBM", 10: "AE: entries from HBM", 11: "AE: backoff scan",
         12: "apply: check by the wave", 13: "apply: record by the wave", 5: "Start", 6: "two-leader check",
         14: "loop passes", 15: "group-ticks run (all groups of the wave)"}
for k, n in extra.items():
    print("%-26s %.4f per wave-tick" % (n, buf[_abi.N_COUNTERS + k] / wt))
