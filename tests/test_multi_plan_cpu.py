"""CPU: the layout arithmetic of the in-process multi-GPU store (ott_multi_plan: what ott_store_reserve plans) — no GPU, no
store: shards tile the rows in order, start on multiples of lcm(chunk size, 8), differ by at most one granule, and coincide with
the chunk-range sharding of the multi-process path (otters_amd.dist.shard_ranges) whenever the chunk size is a multiple of 8."""
import ctypes as C
import math

import numpy as np

from otters_amd import _native as N
from otters_amd.dist import shard_ranges


def plan(n, cs, g):
    out = (C.c_uint64 * g)()
    N.check(N.lib().ott_multi_plan(n, cs, g, out))
    return [int(x) for x in out]


def test_multi_plan_properties():
    rng = np.random.default_rng(0)
    for _ in range(400):
        n = int(rng.choice([0, 1, 7, 100, 4096, 10_000_000, int(rng.integers(1, 2 ** 33))]))
        cs = int(rng.choice([1, 2, 3, 7, 8, 12, 64, 1000, 1024, 4096, 4097]))
        g = int(rng.integers(1, 17))
        st = plan(n, cs, g)
        gran = cs * 8 // math.gcd(cs, 8)
        assert st[0] == 0 and all(a <= b for a, b in zip(st, st[1:])) and st[-1] <= n
        assert all(x % gran == 0 or x == n for x in st)
        sizes = [b - a for a, b in zip(st, st[1:] + [n])]
        assert sum(sizes) == n
        full = [x for x in sizes]
        if n >= gran * g:  # enough granules for everybody: shard sizes differ by at most one granule (the last one by the ragged tail too)
            assert max(full[:-1] or [0]) - min(full[:-1] or [0]) <= gran
            assert full[-1] <= max(full[:-1] or [full[-1]]) + gran
        if cs % 8 == 0:  # the same chunk-range split as the multi-process path
            assert [(a, b - a) for a, b in zip(st, st[1:] + [n])] == shard_ranges(n, cs, g)


def test_multi_plan_keeps_small_stores_on_few_shards(monkeypatch):
    """option / environment multi_min_shard_rows (default 32768; the suite runs with 0): a shard is brought in per that many
    rows, the others stay empty at the end"""
    monkeypatch.setenv("OTT_MULTI_MIN_SHARD_ROWS", "32768")
    assert plan(10_000, 1024, 8) == [0] + [10_000] * 7
    assert plan(65_535, 1024, 8) == [0] + [65_535] * 7
    st = plan(70_000, 1024, 8)  # two shards
    assert st[0] == 0 and st[1] % 1024 == 0 and 33 * 1024 <= st[1] <= 36 * 1024 and st[2:] == [70_000] * 6
    st = plan(200_000, 1024, 8)  # six shards
    assert len(set(st[:6])) == 6 and st[6:] == [200_000] * 2
    st = plan(10_000_000, 1024, 8)
    assert len(set(st)) == 8 and st[-1] < 10_000_000
    monkeypatch.setenv("OTT_MULTI_MIN_SHARD_ROWS", "0")
    assert len(set(plan(10_000, 1024, 8))) == 8


def test_multi_plan_rejects_bad_arguments():
    assert N.lib().ott_multi_plan(10, 8, 0, (C.c_uint64 * 1)()) != 0
    assert N.lib().ott_multi_plan(10, 8, 2, None) != 0
