"""The CPU-side C++ (YAML / .obj parsers, scene assembly, PNG writer, threaded BVH builder) under AddressSanitizer +
UBSan and under ThreadSanitizer: tests/cpp/host_selftest.cpp, built and run by tests/cpp/Makefile. The reference gets
memory and data-race safety from the Rust compiler; this is what the C++ restatement gets instead.
GPU sanitizers are not available on the GPU pool; none of this code needs a GPU."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("target", ["asan", "tsan"])
def test_host_cpp_under_sanitizer(target, tmp_path):
    r = subprocess.run(["make", "-C", str(ROOT / "tests" / "cpp"), target, f"OUT={tmp_path}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "host_selftest ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr and "runtime error" not in r.stderr
