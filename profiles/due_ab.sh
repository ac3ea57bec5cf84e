#!/bin/bash
# A/B of library builds on short launches.  usage: profiles/due_ab.sh <variant> ...  ("base": raft_amd/lib/libraftgpu.so; "<v>+spec": on kernels compiled for the configuration)
for v in "$@"; do
  unset SPEC
  lib=${v%+spec}
  [ "$lib" != "$v" ] && export SPEC=1
  if [ $lib = base ]; then unset RAFTGPU_LIB; else export RAFTGPU_LIB=$PWD/raft_amd/lib/ab/libraftgpu_$lib.so; fi
  echo "== $v"
  NS=${NS:-1,2,4} timeout -k 10 200 python profiles/time_short_launches.py 2>&1 | grep "due launches"
done
