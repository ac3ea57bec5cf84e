cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_try
rm -rf $out; mkdir -p $out
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pass4 -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/pass4.log 2>&1 || { echo fail; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/pass3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/pass3.log 2>&1 || { echo fail; exit 1; }
python profiles/pmc_summary.py $out | grep -E "SIZE|TCC"
