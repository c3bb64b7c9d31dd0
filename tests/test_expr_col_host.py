"""CPU: host mirrors of the expression DSL and Column storage (structural pins from
tests/expr_tests.rs and tests/column_tests.rs; these sit beside the hot path)."""
import numpy as np
import pytest

from otters_amd import Column, ColumnError, DataType, ExprError, col
from otters_amd.col import parse_datetime_millis
from otters_amd.expr import CmpOp

SCHEMA = {"age": DataType.Int32, "score": DataType.Float64, "name": DataType.String, "ts": DataType.DateTime, "w": DataType.Float32}


def leaf(f):
    return (f.kind, f.column, f.cmp, f.rhs if isinstance(f.rhs, str) else (f.rhs.kind, f.rhs.value))


def test_and_or_lowering_to_cnf():
    # (A | B) & C  -> [[A, B], [C]]   (src/expr.rs:345-372)
    e = (col("age").lt(18) | col("age").gt(65)) & col("name").neq("alice")
    cl = e.compile(SCHEMA).clauses
    assert [[leaf(f) for f in c] for c in cl] == [
        [("Numeric", "age", CmpOp.Lt, ("I64", 18)), ("Numeric", "age", CmpOp.Gt, ("I64", 65))],
        [("String", "name", CmpOp.Neq, "alice")]]
    # (A1 & A2) | (B1 & B2) -> cross product (src/expr.rs:485-511)
    e = (col("age").gt(1) & col("age").lt(5)) | (col("score").gte(2) & col("score").lte(3.5))
    cl = e.compile(SCHEMA).clauses
    assert len(cl) == 4 and all(len(c) == 2 for c in cl)
    assert leaf(cl[0][1]) == ("Numeric", "score", CmpOp.Gte, ("F64", 2.0))  # ints widen for float columns


def test_tautology_dropped():
    e = (col("age").eq(3) | col("age").neq(3)) & col("score").gt(1.0)
    assert len(e.compile(SCHEMA).clauses) == 1


def test_datetime_literal_and_errors():
    f = col("ts").gte("2024-01-01T00:00:00Z").compile(SCHEMA).clauses[0][0]
    assert f.rhs.value == 1704067200000 and f.rhs.kind == "I64"
    assert parse_datetime_millis("2024-01-01") == 1704067200000
    assert parse_datetime_millis("2024-01-01 00:00:01") == 1704067201000
    assert parse_datetime_millis("2024-01-01T02:00:00+02:00") == 1704067200000
    assert parse_datetime_millis("2024-01-01T00:00:00.250Z") == 1704067200250
    assert parse_datetime_millis("not a date") is None
    for e, msg in ((col("nope").eq(1), "Unknown column 'nope'"),
                   (col("name").lt("x"), "Unsupported comparator for string column 'name'"),
                   (col("age").eq(1.5), "Type mismatch for column 'age': expected Int32, got literal float"),
                   (col("age").eq("x"), "Type mismatch for column 'age': expected Int32, got literal string"),
                   (col("ts").eq(5), "Type mismatch for column 'ts': expected DateTime, got literal datetime string"),
                   (col("ts").eq("junk"), "Type mismatch for column 'ts': expected DateTime, got literal datetime string"),
                   (col("score").eq("x"), "Type mismatch for column 'score': expected Float64, got literal string"),
                   (col("age"), "Invalid expression (unexpected literal or column without comparator)")):
        with pytest.raises(ExprError) as ei:
            e.compile(SCHEMA)
        assert str(ei.value) == msg


def test_column_storage_and_sentinels():
    c = Column("a", DataType.Int32).from_([1, None, 3])
    assert c.len() == 3 and c.null_mask().tolist() == [False, True, False]
    assert c.i32_values().tolist() == [1, -2147483648, 3] and c.f64_values() is None  # src/col.rs:238-250
    f = Column("f", DataType.Float32).from_([1.5, None])
    assert np.isnan(f.f32_values()[1]) and f.get(1) is None and f.get(0) == 1.5
    s = Column("s", DataType.String).from_(["x", None])
    assert s.string_values() == ["x", ""] and s.null_mask().tolist() == [False, True]
    d = Column("d", DataType.DateTime).from_(["2024-01-05", None, 1704067200000])
    assert d.datetime_values().tolist() == [1704412800000, -2**63, 1704067200000]
    d2 = Column("d", DataType.DateTime).with_datetime_fmt("%d/%m/%Y").from_(["05/01/2024"])
    assert d2.datetime_values().tolist() == [1704412800000]
    with pytest.raises(ColumnError):
        Column("a", DataType.Int32).push("nope")
    with pytest.raises(ColumnError):
        Column("d", DataType.DateTime).push("garbage")
    t = c.take([2, 1])
    assert t.i32_values().tolist() == [3, -2147483648] and t.null_mask().tolist() == [False, True]
