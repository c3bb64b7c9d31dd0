"""GPU: the header-only C++ host mirrors (include/otters.hpp, include/otters_meta.hpp) driving
libotters_hip.so from compiled host code, on the reference's VecStore / MetaStore / zonemap tests
and the README example."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["test_otters_hpp", "test_otters_meta"])
def test_cpp_mirror_binary(name):
    exe = os.path.join(ROOT, "tests", "cpp", name)
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "ALL PASSED" in p.stdout, p.stdout + p.stderr
