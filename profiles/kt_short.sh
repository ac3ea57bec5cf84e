#!/bin/bash
# rocprofv3 kernel trace of one-tick launches on the benchmark workload (kernel durations without the event markers around them).
# usage: profiles/kt_short.sh  -> gpurun_out/kt/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
NS=1 SPAN=500 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 profiles/time_short_launches.py > gpurun_out/kt.log 2>&1
find gpurun_out/kt -name '*kernel_stats.csv' | head -1 | xargs cat | head -8
tail -3 gpurun_out/kt.log
