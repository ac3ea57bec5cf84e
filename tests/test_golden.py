"""Committed vectors (tests/golden/): the CPU restatement, and under -m gpu the HIP engine, must reproduce
them.  The vectors were written by tests/golden/make_golden.py from the restatement itself (the reference has
no fixtures for this path), so they pin the lockstep model against silent change, not against the Go code."""
import json
import os

import pytest

from tests.golden import make_golden

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(make_golden.__file__)), "lockstep_digests.json")))


def test_every_case_has_a_vector():
    assert sorted(GOLDEN) == sorted(make_golden.CASES)


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(name):
    from oracle.binding import OracleEngine
    assert make_golden.run_case(OracleEngine, name) == GOLDEN[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_engine_reproduces_golden(name):
    from raft_amd import Engine
    assert make_golden.run_case(Engine, name) == GOLDEN[name]
