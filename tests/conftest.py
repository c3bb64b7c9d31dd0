import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The in-process multi-GPU store keeps stores of fewer than 2 x 32768 rows on one GPU (option multi_min_shard_rows).  The tests WANT
# small stores spread over every shard (that is how they exercise the sliced masks, the exchange, the merge and the row moves
# on a few thousand rows), so the suite — child processes included — runs with the threshold off; the policy itself has its own
# tests, which set the option explicitly.
os.environ.setdefault("OTT_MULTI_MIN_SHARD_ROWS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import ctypes as C
        from otters_amd import _native as N
        n = C.c_int(0)
        return N.lib().ott_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:  # noqa: BLE001 -- a missing library is reported by the tests that need it
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a CPU-only builder stays green: tests marked `gpu` are skipped there (on the GPU box they run).
    A GPU run must not turn green by skipping everything, though: when the box HAS a GPU device node (/dev/kfd) and the GPU
    tests were asked for (`-m gpu`), or OTT_REQUIRE_GPU=1 is set, a library that cannot see a device is an ERROR — the run
    stops with a non-zero exit code instead of reporting N skipped."""
    if not any(item.get_closest_marker("gpu") for item in items) or _have_gpu():
        return
    markexpr = (config.getoption("markexpr", "") or "").replace(" ", "")
    wants_gpu = "gpu" in markexpr and "notgpu" not in markexpr
    if os.environ.get("OTT_REQUIRE_GPU") == "1" or (wants_gpu and os.path.exists("/dev/kfd")):
        raise pytest.UsageError("GPU tests were requested but libotters_hip.so reports no usable HIP device "
                                "(ott_device_count failed or returned 0; check the build, HIP_VISIBLE_DEVICES and /dev/kfd permissions)")
    skip = pytest.mark.skip(reason="needs an MI355X (no /dev/kfd or no HIP device here)")
    for item in items:
        if item.get_closest_marker("gpu"):
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


# Exchange modes of the in-process multi-GPU store (tests/test_gpu_multi.py's header).  Every test of the multi-store modules runs
# once per mode: "local" and "remote" in this process; "fake_rccl" only in a child process that has OTT_RCCL_LIBRARY set from its
# start (tests/test_gpu_multi_modes.py), because the library binds RCCL once per process.
_MODES = [os.environ["OTT_TEST_MULTI_MODE"]] if os.environ.get("OTT_TEST_MULTI_MODE") else ["local", "remote"]


@pytest.fixture(params=_MODES)
def exchange_mode(request, monkeypatch):
    mode = request.param
    assert mode in ("local", "remote", "fake_rccl"), mode
    for var in ("OTT_MULTI_FAKE_DISTINCT", "OTT_MULTI_TRANSPORT"):
        monkeypatch.delenv(var, raising=False)
    if mode != "local":
        monkeypatch.setenv("OTT_MULTI_FAKE_DISTINCT", "1")  # read by ott_store_create_multi: every shard a device of its own
    if mode == "remote":
        monkeypatch.setenv("OTT_MULTI_TRANSPORT", "1")      # peer copies (the automatic choice would try RCCL first)
    if mode == "fake_rccl":
        from helpers import FAKE_RCCL
        assert os.environ.get("OTT_RCCL_LIBRARY") == FAKE_RCCL, "fake_rccl mode: OTT_RCCL_LIBRARY must be set before the library first looks for RCCL"
    return mode
