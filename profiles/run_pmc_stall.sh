cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_stall
rm -rf $out; mkdir -p $out
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH --output-format csv -d $out/pass1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/pass1.log 2>&1 || { echo fail; tail -3 $out/pass1.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_CYCLES --output-format csv -d $out/pass2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/pass2.log 2>&1 || { echo fail; tail -3 $out/pass2.log; exit 1; }
python profiles/pmc_summary.py $out | grep -E "SQ_"
