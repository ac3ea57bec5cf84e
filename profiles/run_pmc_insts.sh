#!/bin/bash
# instruction counts of one k_tick launch of the bench workload (one PMC pass).  usage: profiles/run_pmc_insts.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_insts
rm -rf $out; mkdir -p $out
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pass1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/pass1.log 2>&1 || { echo fail; tail -3 $out/pass1.log; exit 1; }
python profiles/pmc_summary.py $out | grep -E "SQ_"
