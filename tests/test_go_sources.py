"""The Go sources (cgo binding, lockstep driver) cannot be compiled here; tests/golden/check_go_sources.py holds
their C calls and constants against include/raftgpu.h and oracle/raft_oracle.c from the texts."""
import os

from tests.golden import check_go_sources as chk


def test_go_sources_agree_with_the_header_and_the_oracle():
    errors = []
    used = chk.check_binding(errors)
    chk.check_lockstep(errors)
    assert not errors, "\n".join(errors)
    assert {"rg_create", "rg_step", "rg_start", "rg_get_state", "rg_kill", "rg_crash", "rg_restart", "rg_drain_applied",
            "rg_snapshot_export", "rg_snapshot_import", "rg_allreduce_summary", "rg_write_state"} <= used


def test_the_checker_sees_what_it_is_for(tmp_path, monkeypatch):
    """A wrong name, a wrong arity, a value where a pointer belongs and an unknown constant are each reported."""
    bad = tmp_path / "bad.go"
    bad.write_text('package raft\nfunc f() {\n'
                   '\tC.rg_no_such_call(en.e)\n'
                   '\tC.rg_kill(en.e, grp.g)\n'
                   '\tC.rg_get_state(en.e, grp.g, C.int32_t(1), C.int32_t(0), &lead)\n'
                   '\tC.rg_step(&x, 1)\n'
                   '\t_ = C.RG_NO_SUCH_CONSTANT\n'
                   '\tC.rg_crash(en.e, grp.g, C.int32_t(i))\n}\n')
    monkeypatch.setattr(chk, "GO_DIR", str(tmp_path))
    errors = []
    chk.check_binding(errors)
    text = "\n".join(errors)
    assert "rg_no_such_call is not declared" in text
    assert "rg_kill takes 3 arguments, called with 2" in text
    assert "rg_get_state argument 4" in text and "is a value, the prototype has a pointer" in text
    assert "RG_NO_SUCH_CONSTANT is not defined" in text
    assert len(errors) == 4                      # rg_step(&x, 1) and rg_crash(...) are fine


def test_header_parsing():
    protos = chk.header_prototypes()
    assert protos["rg_start"] == [True, False, False, False, True, True, True]
    assert protos["rg_allreduce_summary"] == [True, True, True, True]
    assert protos["rg_device_count"] == []
    consts = chk.header_constants()
    assert consts["RG_C_AE_ENTRIES_SENT"] == consts["RG_N_COUNTERS"] - 1 == 18
    assert consts["RG_S_FLAGGED_GROUPS"] == 7 and consts["RG_N_SUMMARY"] == 12
    assert consts["RG_OP_KILL"] == 13 and consts["RG_FLAG_REF_PANIC"] == 0x20
    assert os.path.exists(chk.LOCKSTEP)
