"""CPU: MetaStore host logic (Expr compile, zonemap prune, row masks) pinned against the
reference's MetaStore tests (tests/meta_tests.rs, tests/meta_zonemap_tests.rs, README example),
with the scoring done by the ORACLE (meta_query = process_chunk + merge)."""
import numpy as np
import pytest

from helpers import build_meta_case, check_expect, check_stats, load, meta_plan_from_case
from otters_amd import Column, DataType, MetaStore, OttersError, col
from otters_amd.expr import CmpOp
from otters_amd.meta import _range_sat, _row_sat

META_CASES = load("meta_cases.json")
MASK_CASES = load("mask_cases.json")
LANE_PAIR_CASES = load("lane_pair_cases.json")
OPS = {"eq": 0, "neq": 1, "lt": 2, "lte": 3, "gt": 4, "gte": 5}


@pytest.mark.parametrize("case", [c for c in META_CASES if "metric" in c], ids=lambda c: c["name"])
@pytest.mark.parametrize("ties", [0, 1], ids=["literal", "canonical"])
def test_meta_cases_with_oracle(oracle, case, ties):
    meta = build_meta_case(case, host_only=True)
    plan = meta_plan_from_case(case, meta)
    rq, chunk_mask, compiled = plan.resolve()
    row_mask = meta.build_row_mask_host(compiled) if compiled is not None else None
    rows = np.asarray(case["vectors"], np.float32)
    hits, stats = oracle.meta_query(rows, case["chunk_size"], rq.queries, rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr,
                                    chunk_mask=chunk_mask, row_mask=row_mask, ties=ties)
    exp = case["expect"]
    check_expect(hits["index"], hits["score"], {k: v for k, v in exp.items() if k != "stats"})
    if "stats" in exp:
        check_stats(stats, exp["stats"])


def test_meta_build_mismatched_column_len_errors():
    case = next(c for c in META_CASES if c.get("build_error"))
    with pytest.raises(OttersError):
        build_meta_case(case, host_only=True)


def test_meta_filter_compile_error_is_deferred():
    meta = MetaStore.from_columns([Column("age", DataType.Int32).from_([1, 2])]).with_vectors([[1.0], [2.0]]).build(_host_only=True)
    plan = meta.query([1.0], 0).meta_filter(col("nope").gt(1)).take(1)  # no raise here (CHANGELOG 0.1.0-alpha2)
    with pytest.raises(OttersError) as ei:
        plan.collect()
    assert str(ei.value) == "meta_filter compile error: Unknown column 'nope'"
    with pytest.raises(OttersError) as ei:
        meta.query([1.0], 0).meta_filter(col("age").gt(1.5)).collect()
    assert "Type mismatch for column 'age': expected Int32, got literal float" in str(ei.value)


@pytest.mark.parametrize("case", MASK_CASES, ids=lambda c: c["name"])
def test_mask_bit_order(oracle, case):
    got = oracle.rows_mask(case["kind"], case["vals"], case.get("nulls"), 0, len(case["vals"]), OPS[case["op"]], case["thr"])
    assert got.astype(int).tolist() == case["expect_bits"]


@pytest.mark.parametrize("case", LANE_PAIR_CASES, ids=lambda c: c["name"])
def test_lane_pair_known_answers(oracle, case):
    """tests/simd_types_tests.rs: each (a[j], b[j]) of the reference's 8-lane compares / min / max, as one row compare /
    one 2-row zone of the oracle and of the host layer's numpy predicates; only the bits the reference asserts are held."""
    kind, npdt = case["kind"], {"i64": np.int64, "f64": np.float64}[case["kind"]]
    a, b = np.asarray(case["a"], npdt), np.asarray(case["b"], npdt)
    if "op" in case:
        op = OPS[case["op"]]
        mask = mask_host = 0
        for j in range(8):
            mask |= int(oracle.rows_mask(kind, a, None, j, 1, op, b[j].item())[0]) << j
            mask_host |= int(_row_sat(a[j:j + 1], CmpOp(op), b[j])[0]) << j
        assert mask == mask_host
        assert mask & case["set"] == case["set"] and mask & case["clear"] == 0, hex(mask)
    else:
        inter = np.stack([a, b], 1).reshape(-1)  # zone j = rows {a[j], b[j]}
        for j in range(8):
            mn, mx, cnt = oracle.zone_stat(kind, inter, None, 2 * j, 2 * j + 2)
            assert (mn, mx, cnt) == (case["min"][j], case["max"][j], 2)


def test_host_masks_match_oracle_restatement(oracle):
    """the numpy zonemap / row predicates of the host layer vs the oracle's restatement of type_utils.rs"""
    rng = np.random.default_rng(0)
    n = 1000
    for kind, npdt in (("i32", np.int32), ("i64", np.int64), ("f32", np.float32), ("f64", np.float64)):
        if kind[0] == "i":
            vals = rng.integers(-50, 50, n).astype(npdt)
        else:
            vals = rng.normal(0, 20, n).astype(npdt)
        nulls = rng.random(n) < 0.1
        for opname, op in OPS.items():
            thr = npdt(7)
            want = oracle.rows_mask(kind, vals, nulls, 100, 777, op, thr.item())
            got = (_row_sat(vals, CmpOp(op), thr) & ~nulls)[100:877]
            assert np.array_equal(got, want), (kind, opname)
        # zonemaps over 37-row chunks
        cs = 37
        nch = (n + cs - 1) // cs
        mn, mx, nn = [], [], []
        for c in range(nch):
            a, b, cnt = oracle.zone_stat(kind, vals, nulls, c * cs, min((c + 1) * cs, n))
            mn.append(a); mx.append(b); nn.append(cnt)
        if kind == "i32":
            mn = [((v + 2**31) % 2**32) - 2**31 for v in mn]; mx = [((v + 2**31) % 2**32) - 2**31 for v in mx]
        column = Column.from_numpy("c", {"i32": DataType.Int32, "i64": DataType.Int64, "f32": DataType.Float32, "f64": DataType.Float64}[kind], vals, nulls)
        meta = MetaStore.from_columns([column]).with_vectors(np.zeros((n, 2), np.float32)).with_chunk_size(cs).build(_host_only=True)
        z = meta._zones["c"]
        with np.errstate(over="ignore"):
            assert np.array_equal(z.min, np.array(mn).astype(npdt)) and np.array_equal(z.max, np.array(mx).astype(npdt))
        assert np.array_equal(z.non_null, np.array(nn, dtype=np.uint64))
        for opname, op in OPS.items():
            thr = npdt(3)
            want = oracle.chunk_mask(kind, z.min, z.max, z.non_null, op, thr.item())
            got = _range_sat(z.min, z.max, CmpOp(op), thr) & (z.non_null > 0)
            assert np.array_equal(got, want), (kind, opname)


def test_readme_tables_render_like_the_reference(oracle, capsys):
    """README.md:116-150 sample output (data fixture tests/golden/readme_output.txt): head(), the result table and the stats table,
    byte for byte below the title lines.  The README's titles ("MetaStore Head \u2022 ...", "Query Results", "Last Query Stats") are
    those of an earlier revision of the reference: src/display.rs at this one prints "MetaStore \u2022 rows=..." (:155-160), no title
    above the results (:164-188) and "Last Meta Query Stats" (:247) — the mirror follows the source."""
    import os
    from helpers import GOLDEN
    from otters_amd.meta import MetaBuildStats, MetaQueryResults, MetaQueryStats
    want = open(os.path.join(GOLDEN, "readme_output.txt"), encoding="utf-8").read().strip().split("\n\n")
    case = next(c for c in META_CASES if c["name"] == "readme_example_8x4")
    meta = build_meta_case(case, host_only=True)
    head = meta.head()
    assert capsys.readouterr().out == head + "\n"  # printed (src/meta.rs:371-374) and returned
    assert head.split("\n", 1)[1] == want[0].split("\n", 1)[1]
    assert head.split("\n", 1)[0] == want[0].split("\n", 1)[0].replace("MetaStore Head", "MetaStore")
    assert meta.head_n(2).count("\n") == 6  # title, rule, header, rule, two rows, rule
    plan = meta_plan_from_case(case, meta)
    rq, chunk_mask, compiled = plan.resolve()
    hits, _ = oracle.meta_query(np.asarray(case["vectors"], np.float32), 4, rq.queries, rq.metric, rq.take, rq.k,
                                chunk_mask=chunk_mask, row_mask=meta.build_row_mask_host(compiled), ties=oracle.TIES_CANONICAL)
    idx = [int(i) for i in hits["index"]]
    res = MetaQueryResults(sorted(meta.schema()), {n: meta.columns()[n].take(idx) for n in meta.schema()}, idx, [float(s) for s in hits["score"]])
    assert str(res) == want[1].split("\n", 1)[1]
    # the stats tables: the README's numbers through format_query_stats (display.rs:221-249); build stats per display.rs:196-219
    if len(want) > 2:
        st = MetaQueryStats(2, 0, 2, 8, 0.002e-3, 0.031e-3, 0.0, 0.032e-3)
        assert st.format().split("\n", 1)[1] == want[2].split("\n", 1)[1] and st.format().startswith("Last Meta Query Stats\n")
    b = MetaBuildStats(8, 4, 2, 0.0012345, 0.0005, 0.002)
    assert b.format() == ("MetaStore Build Stats\n+------------------+-------+\n| metric           | value |\n+------------------+-------+\n"
                          "| rows             | 8     |\n| dimensions       | 4     |\n| chunks           | 2     |\n| vector_ingest_ms | 1.234 |\n"
                          "| zonemap_build_ms | 0.500 |\n| build_total_ms   | 2.000 |\n+------------------+-------+")
    capsys.readouterr()
    meta.print_last_stats()  # src/meta.rs:562-566: build stats, then the last query's (none yet on this host-only store)
    out = capsys.readouterr().out
    assert out.endswith("(no query stats)\n") and (out.startswith("MetaStore Build Stats\n") or out.startswith("(no build stats)\n"))


def test_row_mask_is_all_true_is_sound_and_fires():
    """The shortcut that skips the row mask claims: every row of every surviving chunk passes.  Whenever it fires, the host row
    mask (src/meta_compute.rs:194-289 restated) must be all ones on the surviving chunks; it must fire for the config-3 filter
    (bucket == chunk parity) and must not for chunks with NULLs, mixed values, float or string leaves."""
    import numpy as np
    from otters_amd import Column, DataType, MetaStore, col
    cs, n = 50, 50 * 40 + 17
    chunk = np.arange(n) // cs
    rng = np.random.default_rng(3)
    bucket = Column.from_numpy("bucket", DataType.Int32, (chunk % 2).astype(np.int32))
    ts = Column.from_numpy("ts", DataType.DateTime, 1_700_000_000_000 + chunk.astype(np.int64) * 86_400_000)
    holes = Column.from_numpy("holes", DataType.Int64, chunk.astype(np.int64), (np.arange(n) % 97) == 0)
    mixed = Column.from_numpy("mixed", DataType.Int32, rng.integers(0, 3, n).astype(np.int32))
    w = Column.from_numpy("w", DataType.Float64, chunk.astype(np.float64))
    g = Column.from_numpy("g", DataType.String, np.array(["a", "b"])[chunk % 2])
    meta = MetaStore.from_columns([bucket, ts, holes, mixed, w, g]).with_vectors(np.ones((n, 4), np.float32)).with_chunk_size(cs).build(_host_only=True)
    fires = {
        "bucket==1": (col("bucket").eq(1), True),
        "bucket!=0": (col("bucket").neq(0), False),                                   # != never prunes: the bucket-0 chunks survive and fail row by row
        "bucket!=7": (col("bucket").neq(7), True),
        "bucket>=1 & ts<2023-12-01": (col("bucket").gte(1) & col("ts").lt("2023-12-01"), True),
        "bucket==1 | mixed==2": (col("bucket").eq(1) | col("mixed").eq(2), False),   # chunks kept through `mixed` are not wholly satisfied
        "mixed<=2": (col("mixed").lte(2), True),
        "mixed<2": (col("mixed").lt(2), False),
        "holes>=3": (col("holes").gte(3), False),                                      # NULL rows fail
        "w>=3 (float)": (col("w").gte(3.0), False),
        "g=='a' (string)": (col("g").eq("a"), False),
    }
    for name, (expr, want) in fires.items():
        compiled = expr.compile(meta.schema())
        cm = meta.build_chunk_mask_for_plan(compiled)
        got = meta.row_mask_is_all_true(compiled, cm)
        assert got == want, name
        if got:
            rows = meta.build_row_mask_host(compiled)
            assert rows[np.repeat(cm, cs)[:n]].all(), name
    # randomised soundness: whenever it fires, the host mask agrees
    for seed in range(200):
        r = np.random.default_rng(seed)
        ops = ["eq", "neq", "lt", "lte", "gt", "gte"]
        def leaf():
            c = r.choice(["bucket", "holes", "mixed"])
            return getattr(col(str(c)), str(r.choice(ops)))(int(r.integers(-1, 42)))
        expr = leaf()
        for _ in range(int(r.integers(0, 3))):
            expr = (expr & leaf()) if r.random() < 0.5 else (expr | leaf())
        compiled = expr.compile(meta.schema())
        cm = meta.build_chunk_mask_for_plan(compiled)
        if meta.row_mask_is_all_true(compiled, cm):
            assert meta.build_row_mask_host(compiled)[np.repeat(cm, cs)[:n]].all(), seed
