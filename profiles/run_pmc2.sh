#!/bin/bash
# Extra PMC passes on one full batch (49,152 groups: exactly 4 waves per SIMD).  usage: profiles/run_pmc2.sh <tag>
set -u
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc2_$tag
mkdir -p $out
export PG=49152
run() {
  n=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pass$n -- python3 profiles/time_ticks.py pmc > $out/pass$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out/pass$n.log; return 1; }
}
run 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU &&
run 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH &&
run 3 SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
python3 profiles/pmc_summary.py $out
