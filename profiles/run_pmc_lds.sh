cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_lds
rm -rf $out; mkdir -p $out
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_LDS_ADDR_CONFLICT --output-format csv -d $out/pass1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/pass1.log 2>&1 || { echo fail; tail -3 $out/pass1.log; exit 1; }
python profiles/pmc_summary.py $out | grep -E "SQ_"
