"""raft_amd: host side of the batched lockstep Raft engine for MI355X.

`Engine` is the handle on the HIP engine (C ABI in include/raftgpu.h);
`raft_amd.raft` mirrors the reference's per-peer API
(reference: src/raft/raft.go:700,344,90,377; src/raft/persister.go:20-70).
"""
from ._abi import Config, make_config  # noqa: F401
from .engine import Engine, RaftGpuError  # noqa: F401
