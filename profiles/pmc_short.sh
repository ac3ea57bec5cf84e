#!/bin/bash
# Issue counters of one-tick due launches on the benchmark workload (one PMC pass per run, no tracing).
# usage: profiles/pmc_short.sh  -> gpurun_out/pmc_short/summary.txt  (sums and per-launch means for k_tick_due)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_short
rm -rf $out && mkdir -p $out
export NS=1 SPAN=300
run() {
  n=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pass$n -- python3 profiles/time_short_launches.py > $out/pass$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out/pass$n.log; return 1; }
}
run 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU &&
run 2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU &&
run 3 GRBM_GUI_ACTIVE GRBM_COUNT &&
python3 profiles/pmc_summary.py $out k_tick_due > $out/summary.txt && cat $out/summary.txt
