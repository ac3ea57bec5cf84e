"""How full the passes of the tick loop are: groups that run a tick per pass, of the groups a wave holds (diagnostic
-DRG_SECTION_COUNTS build).  usage: RAFTGPU_LIB=$PWD/raft_amd/lib/libraftgpu_stamps.so python profiles/pass_fill.py [ticks]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import Engine, _abi
G, T = int(os.environ.get("PG", 65536)), int(sys.argv[1]) if len(sys.argv) > 1 else 20000
kw = dict(n_groups=G, n_peers=5, log_cap=1024, seed=2, req_drop_permille=100, reply_drop_permille=100, start_period=5, start_max=1000, part_period=100, part_jitter=20)
e = Engine(_abi.make_config(**kw))
e.step(T); e.sync()
buf = (C.c_uint64 * 56)()
e._lib.rg_debug_section_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert e._lib.rg_debug_section_counts(e._h, buf) == 0
passes, runs = buf[_abi.N_COUNTERS + 14], buf[_abi.N_COUNTERS + 15]
print("chunks=%s: %d passes, %d group event-ticks = %.3f per group-tick, %.2f groups per pass (of 12), kernel %.2f ms" % (
    os.environ.get("RAFTGPU_RELAY_CHUNKS", "default"), passes, runs, runs / (G * T), runs / passes, e.step_time_ms()[0]))
