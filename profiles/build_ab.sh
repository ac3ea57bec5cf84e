#!/bin/bash
# A/B build of the engine for five-peer groups only (a tenth of the full compile time).
# usage: profiles/build_ab.sh <variant> [extra compiler flags]  -> raft_amd/lib/ab/libraftgpu_<variant>.so  (RAFTGPU_LIB selects it)
set -e
v=$1; shift
cd "$(dirname "$0")/.."
mkdir -p raft_amd/lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -w -I include -I raft_amd/csrc -ldl -DRG_ONLY_P=5 "$@" \
  -o raft_amd/lib/ab/libraftgpu_$v.so raft_amd/csrc/raftgpu.hip
echo built raft_amd/lib/ab/libraftgpu_$v.so
