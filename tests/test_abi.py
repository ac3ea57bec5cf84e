"""The C-ABI library loads and exports every symbol include/raftgpu.h declares.
No compute calls here: this runs without a GPU."""
import ctypes
import os
import re

import pytest

from raft_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    """What the header declares for the shipped library (the block for diagnostic builds left out)."""
    text = open(os.path.join(ROOT, "include", "raftgpu.h")).read()
    text = re.sub(r"#if defined\(RG_SECTION_COUNTS\).*?#endif", "", text, flags=re.S)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z_0-9]+)\s*\(", text)))


def test_header_and_loader_agree():
    assert _declared() == sorted(_abi.EXPORTS)


def test_library_exports_exactly_what_the_header_declares():
    """Both ways, from the library's own dynamic symbol table: nothing declared is missing, nothing exported is
    undeclared."""
    import subprocess
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TW" and ln.split()[-1].startswith("rg_"))
    assert exported == _declared()


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_abi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_checker_header_and_loader_agree():
    from raft_amd import lincheck
    text = open(os.path.join(ROOT, "include", "lincheck.h")).read()
    declared = sorted(set(re.findall(r"\b(lc_[a-z_0-9]+)\s*\(", text)))
    assert declared == sorted(lincheck.EXPORTS)
    lib = ctypes.CDLL(lincheck.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_strerror_without_gpu():
    lib = _abi.load_library()
    assert b"gfx950" in lib.rg_version()
    assert lib.rg_strerror(0) == b"ok"
    assert lib.rg_strerror(_abi.ERR_HIP) != lib.rg_strerror(_abi.ERR_ARG)


def test_config_struct_layout_matches_header():
    assert ctypes.sizeof(_abi.Config) == 4 * 4 + 8 + 8 + 4 * 3 + 4 * 4 + 4 * 2 + 4 * 4 + 4 * 3
    assert ctypes.sizeof(_abi.AppendArgs) == 32 and ctypes.sizeof(_abi.StartReq) == 32
    assert ctypes.sizeof(_abi.VoteArgs) == 16 and ctypes.sizeof(_abi.ApplyMsgC) == 16


def test_engine_fails_loudly_without_a_gpu():
    """No CPU fallback: on a machine without a gfx950 device creation must fail, not degrade."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raft_amd import Engine, RaftGpuError
    with pytest.raises(RaftGpuError):
        Engine(_abi.make_config(n_groups=1, n_peers=3, log_cap=8))


def test_oracle_is_not_imported_by_the_product():
    for root, _, files in os.walk(os.path.join(ROOT, "raft_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("# oracle", ""), os.path.join(root, f)


def test_library_embeds_its_device_sources():
    """rg_specialize compiles the sources the library carries: they have to be the ones it was built from."""
    import __graft_entry__
    __graft_entry__.build()                 # rebuilds only what is older than its sources
    blob = open(_abi.LIB_PATH, "rb").read()
    for rel in ("include/raftgpu.h", "raft_amd/csrc/rg_state.h", "raft_amd/csrc/rg_tick.h"):
        assert open(os.path.join(ROOT, rel), "rb").read() in blob, "%s is not what libraftgpu.so embeds: rebuild" % rel
