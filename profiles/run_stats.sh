#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command.  usage: profiles/run_stats.sh <tag> -> gpurun_out/prof_<tag>/
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $out.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $out.log; exit 1; }
tail -1 $out.log
find $out -name '*kernel_stats.csv' | head -1 | xargs cat
