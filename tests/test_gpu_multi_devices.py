"""GPU: ONE store over several DISTINCT GPUs of this process — tests/multi_devices_check.py in a child process.

On a machine with two GPUs or more it runs over the first two and over all of them, with the peer-copy transport, the RCCL
transport (ncclCommInitAll + grouped ncclAllGather) and the automatic choice: the first execution of ott_multi.hip's
cross-device paths over xGMI (the development boxes have one GPU, where tests/test_gpu_multi.py repeats ordinal 0).  On a
one-GPU box the same script runs with the list 0,0, so the script itself cannot rot."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from otters_amd import _native as N

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "multi_devices_check.py")


def _n_devices() -> int:
    n = C.c_int(0)
    N.check(N.lib().ott_device_count(C.byref(n)))
    return n.value


def _run(args, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, SCRIPT] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT, stdin=subprocess.DEVNULL)
    assert p.returncode == 0 and "ALL OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_check_script_on_repeated_ordinals():
    out = _run(["--devices", "0,0"], 300)
    assert "OK devices [0, 0] transport peer -> peer" in out


def test_distinct_devices_peer_rccl_auto():
    n = _n_devices()
    if n < 2:
        pytest.skip(f"this machine has {n} GPU: distinct device ordinals (xGMI) cannot be exercised here")
    out = _run([], 900)
    assert f"OK devices {list(range(n))} transport peer -> peer" in out
    assert f"OK devices {list(range(n))} transport rccl -> rccl" in out
