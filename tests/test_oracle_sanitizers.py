"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no sanitizer story:
it is single-threaded Python; the restatement in C gets one)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_selftest_under_asan_ubsan():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
