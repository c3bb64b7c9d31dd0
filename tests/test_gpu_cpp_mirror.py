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


def test_plain_c_host_readme_example():
    """tests/c/c_host.c: a pure-C11 program (what a Rust extern "C" binding amounts to) runs the reference's README example
    through ott_query and ott_query_sharded; its output must be the reference's documented result (README.md:129-150)."""
    exe = os.path.join(ROOT, "tests", "c", "c_host")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[0] == "chunks 2 evaluated 2 compared 8"
    assert lines[1:] == ["hit 4 score 0.970142", "hit 2 score 0.707107", "hit 6 score 0.707107"]
