"""CPU: the reference's own host-planner suites — tests/expr_tests.rs and tests/column_tests.rs, transcribed as data by
tests/golden/make_host_golden.py — run against the Python mirror (otters_amd/expr.py, col.py) and, through
tests/cpp/test_host_golden (built from tests/cpp/test_host_golden.cpp), against the C++ mirror (include/otters_meta.hpp)."""
import contextlib
import io
import json
import math
import os
import subprocess

import numpy as np
import pytest

from otters_amd import Column, ColumnError, DataType, ExprError, col
from otters_amd.expr import CmpOp, Expr, lit

HERE = os.path.dirname(os.path.abspath(__file__))
EXPR = json.load(open(os.path.join(HERE, "golden", "expr_cases.json")))
COLS = json.load(open(os.path.join(HERE, "golden", "column_cases.json")))
OPS = {"eq": CmpOp.Eq, "neq": CmpOp.Neq, "lt": CmpOp.Lt, "lte": CmpOp.Lte, "gt": CmpOp.Gt, "gte": CmpOp.Gte}
SCHEMA = {k: DataType[v] for k, v in EXPR["schema"].items()}


def literal(d):
    if d is None:
        return None
    (tag, v), = d.items()
    return {"i": int, "f": float, "s": str}[tag](v)


def build(e) -> Expr:
    if "cmp" in e:
        c, op, l = e["cmp"]
        return getattr(col(c), op)(literal(l))
    if "and" in e:
        return build(e["and"][0]) & build(e["and"][1])
    if "or" in e:
        return build(e["or"][0]) | build(e["or"][1])
    if "col" in e:
        return col(e["col"])
    if "lit" in e:
        return lit(literal(e["lit"]))
    r = e["raw_cmp"]  # Expr::Cmp built by hand (tests/expr_tests.rs:36-40)
    return Expr("Cmp", build(r["left"]), build(r["right"]), OPS[r["op"]])


def leaf(f):
    d = {"kind": f.kind, "column": f.column, "cmp": f.cmp.name.lower()}
    d["rhs"] = f.rhs if isinstance(f.rhs, str) else {f.rhs.kind: f.rhs.value}
    return d


@pytest.mark.parametrize("case", EXPR["cases"], ids=[c["name"] for c in EXPR["cases"]])
def test_expr_case(case):
    want = case["expect"]
    if "error" in want:
        with pytest.raises(ExprError) as ei:
            build(case["expr"]).compile(SCHEMA)
        assert str(ei.value) == want["display"], case["ref"]
        return
    cl = build(case["expr"]).compile(SCHEMA).clauses
    got = [[leaf(f) for f in c] for c in cl]
    if "clauses" in want:
        assert got == want["clauses"], case["ref"]
        for c, wc in zip(got, want["clauses"]):  # I64 stays an int, F64 a float (NumericLiteral's variant)
            for f, wf in zip(c, wc):
                if isinstance(wf["rhs"], dict):
                    (k, v), = f["rhs"].items()
                    assert isinstance(v, float if k == "F64" else int), (case["ref"], f)
    if "n_clauses" in want:
        assert len(got) == want["n_clauses"], case["ref"]
    if "clause_sizes" in want:
        assert [len(c) for c in got] == want["clause_sizes"], case["ref"]
    if "clause_sizes_sorted" in want:
        assert sorted(len(c) for c in got) == want["clause_sizes_sorted"], case["ref"]
    if "first_leaf_kinds" in want:
        assert [c[0]["kind"] for c in got] == want["first_leaf_kinds"], case["ref"]


def check_column(c: Column, want, ref):
    for key, v in want.items():
        if key == "name":
            assert c.name() == v, ref
        elif key == "dtype":
            assert c.dtype() == DataType[v], ref
        elif key == "len":
            assert c.len() == v, ref
        elif key == "is_empty":
            assert c.is_empty() == v, ref
        elif key == "null_mask":
            assert c.null_mask().tolist() == v, ref  # true = NULL, row order (BitVec Lsb0: bit i = row i)
        elif key == "accessors":
            for acc, n in v.items():
                got = getattr(c, acc + "_values")()
                assert (got is None) if n is None else (got is not None and len(got) == n), (ref, acc)
        elif key == "values_len":
            assert len(c.values()) == v, ref
        elif key == "values_is_empty":
            assert (len(c.values()) == 0) == v, ref
        elif key == "values_dtype":
            assert c.data_type() == DataType[v], ref
        elif key == "raw":
            got = list(c.values()) if c.dtype() == DataType.String else c.values().tolist()
            assert len(got) == len(v), ref
            for g, w in zip(got, v):
                assert (isinstance(g, float) and math.isnan(g)) if w == "NaN" else g == w, (ref, got, v)
        elif key == "head_n":
            with contextlib.redirect_stdout(io.StringIO()) as out:
                text = c.head_n(v[0])
            assert text == v[1] and out.getvalue() == v[1] + "\n", (ref, text)
        else:
            raise AssertionError(f"unknown expectation {key}")


@pytest.mark.parametrize("case", COLS, ids=[c["name"] for c in COLS])
def test_column_case(case):
    c = None
    for step in case["steps"]:
        if "new" in step:
            n = step["new"]
            c = Column.new(n["name"], DataType[n["dtype"]])
            if n["fmt"]:
                c = c.with_datetime_fmt(n["fmt"])
        elif "push" in step or "from" in step or "from_range" in step:
            def run():
                if "push" in step:
                    c.push(literal(step["push"]))
                elif "from" in step:
                    assert c.from_([literal(v) for v in step["from"]]) is c
                else:
                    assert c.from_(range(*step["from_range"])) is c
            if step["ok"]:
                run()
            else:
                with pytest.raises(ColumnError) as ei:
                    run()
                if step.get("error") == "ParseError":
                    assert str(ei.value).startswith("Parse error: "), case["ref"]  # ColumnError::ParseError's Display, src/col.rs:86-97
        else:
            check_column(c, step["expect"], case["ref"])


def test_fixtures_cover_every_reference_test():
    """15 #[test] functions in tests/expr_tests.rs, 22 in tests/column_tests.rs: every one has at least one case citing its lines"""
    e = {c["ref"].split(":")[1].split("-")[0] for c in EXPR["cases"] if c["ref"].startswith("tests/expr_tests.rs")}
    assert len(e) == 16  # type_mismatch_errs (:93-103) is two cases, one per assertion
    k = [c for c in COLS if c["ref"].startswith("tests/column_tests.rs")]
    assert len(k) == 24  # test_values_method (:318-346) is three cases, one per column type
    assert len({c["name"].rsplit("_int32", 1)[0].rsplit("_float64", 1)[0].rsplit("_string", 1)[0] if c["name"].startswith("test_values_method") else c["name"] for c in k}) == 22


def test_cpp_mirror_passes_the_same_fixtures():
    """the C++ mirror against the SAME two files (host-only code: runs without a GPU)"""
    exe = os.path.join(HERE, "cpp", "test_host_golden")
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "-s", "test_host_golden"])
    out = subprocess.run([exe, os.path.join(HERE, "golden", "expr_cases.json"), os.path.join(HERE, "golden", "column_cases.json")],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout, out.stdout + out.stderr
    assert f"{len(EXPR['cases'])} expr cases, {len(COLS)} column cases" in out.stdout, out.stdout
