#!/bin/bash
# Two more PMC passes over one launch of the bench workload: instruction fetch / branches / scalar and LDS pipe, and the
# instruction cache.  usage: profiles/run_pmc_extra.sh <tag>  -> gpurun_out/pmcx_<tag>/summary.txt
set -u
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcx_$tag
rm -rf $out && mkdir -p $out
run() {
  n=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pass$n -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/pass$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out/pass$n.log; return 1; }
}
run 1 SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run 2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE &&
python3 profiles/pmc_summary.py $out k_tick > $out/summary.txt && cat $out/summary.txt
