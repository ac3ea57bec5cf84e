#!/bin/bash
# Bench workload with extra compiler options for the specialised kernel (RAFTGPU_SPEC_DEFS), one run per argument.
# usage: profiles/spec_sweep.sh <tag> "<opts>" "<opts>" ...     ("" = none) -> gpurun_out/<tag>_<n>.json
tag=$1; shift
n=0
for o in "$@"; do
  n=$((n + 1))
  RAFTGPU_SPEC_DEFS="$o" timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err || { echo "'$o' failed"; tail -3 gpurun_out/${tag}_$n.err; exit 1; }
  python - "$o" gpurun_out/${tag}_$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
c = d["counters_per_step"]
print("%-60s %.4e group-ticks/s  %s %.2f ms  committed %d  flagged %d" % (sys.argv[1] or "(none)", d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], c["entries_committed"], d["flagged_groups"]))
PY
done
