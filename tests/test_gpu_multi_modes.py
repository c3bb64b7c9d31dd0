"""GPU: the cross-device branches of the in-process multi-GPU store (otters_amd/csrc/ott_multi.hip), run and AUDITED on the
one-GPU box (src/meta.rs:678-709 is what they replace: fan-out over chunks, concat, sort, truncate).

tests/test_gpu_multi.py and tests/test_gpu_multi_fuzz.py hold the store to ONE single-GPU store bit for bit in the exchange modes
"local" and "remote" inside the main pytest process.  Here the same two files run in child processes
  * over tests/fake_rccl (mode "fake_rccl": the grouped ncclAllGather branch with G = 2 .. 8 ranks; the RCCL binding is chosen once
    per process, hence the child), and
  * under the DEVICE-AFFINITY AUDIT build of the library (libotters_hip_audit.so, `make -C otters_amd/csrc audit`): every HIP call
    the library makes is checked against the logical device its thread selected — a missed use_device() on a shard thread, in the
    background plane builder or in drain() aborts the child — in all three modes.
The audit build checks itself first (three deliberate mistakes must be noticed)."""
import os
import subprocess
import sys

import pytest

from helpers import FAKE_RCCL

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT_LIB = os.path.join(ROOT, "otters_amd", "csrc", "libotters_hip_audit.so")
FILES = ["tests/test_gpu_multi.py", "tests/test_gpu_multi_fuzz.py"]


def child_env(mode, audit):
    env = dict(os.environ, OTT_TEST_MULTI_MODE=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("OTT_MULTI_FAKE_DISTINCT", "OTT_MULTI_TRANSPORT", "OTT_RCCL_LIBRARY", "OTT_TEST_HOOKS", "OTT_LIB_PATH"):
        env.pop(var, None)
    if mode == "fake_rccl":
        assert os.path.exists(FAKE_RCCL), "tests/fake_rccl/libfake_rccl.so is missing: __graft_entry__.build() makes it"
        env["OTT_RCCL_LIBRARY"] = FAKE_RCCL
        env["OTT_TEST_HOOKS"] = "1"  # (the library ignores OTT_RCCL_LIBRARY without it)
    if audit:
        assert os.path.exists(AUDIT_LIB), "libotters_hip_audit.so is missing: __graft_entry__.build() makes it"
        env["OTT_LIB_PATH"] = AUDIT_LIB
    return env


def test_the_audit_build_notices_deliberate_mistakes():
    code = ("import ctypes as C, os\n"
            f"L = C.CDLL({AUDIT_LIB!r})\n"
            "assert L.ott_audit_selftest(0) == 3, 'the audit missed a deliberate mistake'\n"
            "assert L.ott_audit_violations() == 0\n"
            "print('selftest ok')\n")
    env = child_env("local", True)
    env["PYTHONPATH"] = ROOT
    pre = "from otters_amd import _native as N\nN._preload_torch_hip()\n"
    out = subprocess.run([sys.executable, "-c", pre + code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "selftest ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
    assert out.stderr.count("OTT_DEVICE_AUDIT violation") == 3, out.stderr[-4000:]


@pytest.mark.parametrize("mode,audit", [("fake_rccl", False), ("local", True), ("remote", True), ("fake_rccl", True)],
                         ids=["fake_rccl", "audit-local", "audit-remote", "audit-fake_rccl"])
def test_multi_store_suite_in_a_child_process(mode, audit):
    env = child_env(mode, audit)
    env.setdefault("OTT_MULTI_FUZZ_SEEDS", "12")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-s", "-m", "gpu", "-p", "no:cacheprovider"] + FILES  # -s: a violation's line reaches the parent
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "OTT_DEVICE_AUDIT violation" not in out.stderr, tail
