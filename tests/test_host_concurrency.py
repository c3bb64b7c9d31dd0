"""CPU: the library's host-side concurrency under the sanitizers (sanitizers on the CPU build).

otters_amd/csrc/ott_host.h holds, HIP-free, what the C++ behind the C ABI needs where the reference has the borrow checker
(`&self` queries src/vec.rs:387, MetaStore !Sync src/meta.rs:54, rayon src/meta.rs:678): the reader / writer lock with writers
first (RwGate), the shard thread pool with its completion latch (ShardPool), the pool of query contexts (ContextPool), rows staged
on the host (StagedRows), "take the store shared only once it is clean" (lock_shared_clean) and the background worker behind the
hi-plane prebuild (QuietWorker).  libotters_hip.so is built from that header; tests/host/host_concurrency.cpp compiles the SAME
header against a mock device (host memory for HBM, reallocated on growth so that a reader racing an append is a use-after-free)
and runs randomised schedules — concurrent run_all callers, pools and workers destroyed idle / mid-run, more threads than
contexts, single-row appends + large appends + queries + the background builder, a front store over shards — under
-fsanitize=thread and -fsanitize=address,undefined.  Any sanitizer report or failed invariant fails the test.

Default: 24 000 schedules under TSan + 12 000 under ASan/UBSan (a few seconds each on 4 processes); OTT_HOST_SCHEDULES scales both
(the round-5 soak: OTT_HOST_SCHEDULES=100000, profiles/round5/host_sanitizers.md)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
TOTAL = int(os.environ.get("OTT_HOST_SCHEDULES", "24000"))
PROCS = 4


@pytest.fixture(scope="module")
def binaries():
    subprocess.check_call(["make", "-C", HERE, "-s"])
    return {k: os.path.join(HERE, k) for k in ("host_tsan", "host_asan")}


def run_many(exe, schedules, extra_env):
    env = dict(os.environ, **extra_env)

    def one(seed):
        return subprocess.run([exe, str(schedules), str(seed)], env=env, capture_output=True, text=True, timeout=3000)
    with ThreadPoolExecutor(PROCS) as ex:
        outs = list(ex.map(one, range(1, PROCS + 1)))
    for seed, out in enumerate(outs, 1):
        tail = (out.stdout[-1500:], out.stderr[-4000:])
        assert out.returncode == 0, (seed, tail)
        assert f"OK schedules={schedules} " in out.stdout, (seed, tail)
        assert "Sanitizer" not in out.stderr and "runtime error" not in out.stderr and "CHECK failed" not in out.stderr, (seed, tail)


def test_host_concurrency_under_thread_sanitizer(binaries):
    run_many(binaries["host_tsan"], TOTAL // PROCS, {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1 exitcode=66"})


def test_host_concurrency_under_address_and_ub_sanitizers(binaries):
    run_many(binaries["host_asan"], TOTAL // (2 * PROCS), {"ASAN_OPTIONS": "detect_leaks=1:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})


def test_every_scenario_kind_alone(binaries):
    """each of the five scenario kinds on its own seed stream (a kind that stops making progress — a writer starved by
    readers did, before RwGate — shows up as a timeout here instead of hiding in the mix)"""
    for kind in range(5):
        out = subprocess.run([binaries["host_tsan"], "300", "11", str(kind)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
        assert out.returncode == 0 and "OK schedules=300 " in out.stdout and "Sanitizer" not in out.stderr, (kind, out.stdout[-500:], out.stderr[-3000:])
