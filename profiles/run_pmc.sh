#!/bin/bash
# PMC passes for k_tick (one pass per run; counters never mixed with tracing).
# usage: profiles/run_pmc.sh <tag> [bench|cfg2|cfg4|cfg5|target]   -> gpurun_out/pmc_<tag>/passN/
# "bench": the default bench command; the others: that configuration of profiles/time_configs.py (2 launches each).
set -u
tag=${1:-r02}
what=${2:-bench}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
if [ $what = bench ]; then cmd="bench.py --steps 1 --warmup 0 --no-cpu"; else export ONLY=$what; cmd="profiles/time_configs.py"; fi
run() {
  n=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pass$n -- python3 $cmd > $out/pass$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out/pass$n.log; return 1; }
}
run 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU &&
run 2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU &&
run 3 FETCH_SIZE TCC_HIT_sum &&
run 4 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum &&
run 5 GRBM_GUI_ACTIVE GRBM_COUNT &&
python3 profiles/pmc_summary.py $out k_tick > $out/summary.txt && cat $out/summary.txt
