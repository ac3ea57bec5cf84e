"""The CPU code under AddressSanitizer and UndefinedBehaviorSanitizer (host only): the checker through every entry
point on lossy, churning and scheduled runs, and the linearizability checker on random histories."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("prog", ["sanitize_oracle", "sanitize_lincheck"])
def test_clean_under_asan_ubsan(prog):
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, prog], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(d, prog)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitize OK" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
