#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RAFTGPU_LIB=$PWD/raft_amd/lib/ab/libraftgpu_blk.so
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
NS=1 SPAN=500 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 profiles/time_short_launches.py > gpurun_out/kt.log 2>&1
find gpurun_out/kt -name '*kernel_stats.csv' | head -1 | xargs cat | head -8
tail -3 gpurun_out/kt.log
