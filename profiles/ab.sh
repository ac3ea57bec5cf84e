#!/bin/bash
# A/B timing of library builds on the bench workload.  usage: profiles/ab.sh <tag> <variant> ...   (variant "base" = raft_amd/lib/libraftgpu.so,
# else raft_amd/lib/ab/libraftgpu_<variant>.so) -> gpurun_out/<tag>_<variant>.json
tag=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset RAFTGPU_LIB; else export RAFTGPU_LIB=$PWD/raft_amd/lib/ab/libraftgpu_$v.so; fi
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/${tag}_$v.json 2> gpurun_out/${tag}_$v.err || { echo "$v failed"; tail -3 gpurun_out/${tag}_$v.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$v.json"))
c = d["counters_per_step"]
print("%-8s %.4e group-ticks/s  kernel %.2f ms  committed %d  rv %d ae %d" % ("$v", d["value"], d["roofline"]["kernel_ms"], c["entries_committed"], c["rv_rounds"], c["ae_rounds"]))
PY
done
