#!/bin/bash
# A/B of compiler options / feature macros of the kernel rg_specialize builds, on the bench workload.
# usage: [RAFTGPU_LIB=...] profiles/spec_ab.sh <tag> "<defs A>" "<defs B>" ...   ("-" = none) -> gpurun_out/<tag>_<i>.json
tag=$1; shift
i=0
for d in "$@"; do
  if [ "$d" = "-" ]; then unset RAFTGPU_SPEC_DEFS; else export RAFTGPU_SPEC_DEFS="$d"; fi
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/${tag}_$i.json 2> gpurun_out/${tag}_$i.err || { echo "$d failed"; tail -3 gpurun_out/${tag}_$i.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$i.json"))
c = d["counters_per_step"]
print("%-40s %.4e group-ticks/s  kernel %.2f ms  committed %d  rv %d ae %d flagged %d" % ("$d", d["value"], d["roofline"]["kernel_ms"], c["entries_committed"], c["rv_rounds"], c["ae_rounds"], d["flagged_groups"]))
PY
  i=$((i+1))
done
