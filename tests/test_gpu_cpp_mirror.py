"""GPU: the header-only C++ host mirrors (include/otters.hpp, include/otters_meta.hpp) driving
libotters_hip.so from compiled host code, on the reference's VecStore / MetaStore / zonemap tests
and the README example."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("devices", [None, "0,0,0"], ids=["one_gpu", "three_shards"])
@pytest.mark.parametrize("name", ["test_otters_hpp", "test_otters_meta"])
def test_cpp_mirror_binary(name, devices):
    """devices = "0,0,0": the SAME binaries with OTTERS_HIP_DEVICES set — every VecStore / MetaStore they build is then one
    store over three shards of the process (ott_store_create_multi); the reference's tests must pass unchanged."""
    exe = os.path.join(ROOT, "tests", "cpp", name)
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    env = dict(os.environ)
    env.pop("OTTERS_HIP_DEVICES", None)
    if devices:
        env["OTTERS_HIP_DEVICES"] = devices
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0 and "ALL PASSED" in p.stdout, p.stdout + p.stderr


def test_plain_c_host_readme_example():
    """tests/c/c_host.c: a pure-C11 program (what a Rust extern "C" binding amounts to) runs the reference's README example
    through ott_query, ott_query_sharded and a multi-GPU store (ott_store_create_multi); its output must be the reference's documented result (README.md:129-150)."""
    exe = os.path.join(ROOT, "tests", "c", "c_host")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[0] == "multi shards 3 transport peer"  # the same example on ONE store over three shards gave the same hits
    assert lines[1] == "chunks 2 evaluated 2 compared 8"
    assert lines[2:] == ["hit 4 score 0.970142", "hit 2 score 0.707107", "hit 6 score 0.707107"]
