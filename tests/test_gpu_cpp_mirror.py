"""GPU: the header-only C++ host mirror (include/otters.hpp) driving libotters_hip.so from
compiled host code, on a subset of the reference's VecStore tests."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_binary():
    exe = os.path.join(ROOT, "tests", "cpp", "test_otters_hpp")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "ALL PASSED" in p.stdout, p.stdout + p.stderr
