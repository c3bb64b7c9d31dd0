"""Host-side mirror of otters' expression DSL (src/expr.rs): `col("age").gt(15) & col("grade").eq("A")`
-> type-checked CNF `CompiledFilter` (AND of clauses, each an OR of column filters)."""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Dict, List, Union

from .col import DataType, parse_datetime_millis


class CmpOp(enum.IntEnum):  # src/expr.rs:83-91 (values = ott_op in the C ABI)
    Eq = 0
    Neq = 1
    Lt = 2
    Lte = 3
    Gt = 4
    Gte = 5


class ExprError(Exception):
    """src/expr.rs:228-263; str(e) is the reference's Display text."""


@dataclass(frozen=True)
class NumericLiteral:  # src/expr.rs:192-196
    kind: str  # "I64" | "F64"
    value: Union[int, float]


@dataclass(frozen=True)
class ColumnFilter:  # src/expr.rs:198-211
    kind: str  # "Numeric" | "String"
    column: str
    cmp: CmpOp
    rhs: Union[NumericLiteral, str]


@dataclass
class CompiledFilter:  # src/expr.rs:222-226
    clauses: List[List[ColumnFilter]]


class Expr:
    """src/expr.rs:93-190.  kind: "Column" | "Literal" | "Cmp" | "And" | "Or"."""

    def __init__(self, kind, a=None, b=None, op=None):
        self.kind, self.a, self.b, self.op = kind, a, b, op

    def _cmp(self, v, op: CmpOp) -> "Expr":
        return Expr("Cmp", self, lit(v), op)

    def eq(self, v) -> "Expr":
        return self._cmp(v, CmpOp.Eq)

    def neq(self, v) -> "Expr":
        return self._cmp(v, CmpOp.Neq)

    def lt(self, v) -> "Expr":
        return self._cmp(v, CmpOp.Lt)

    def lte(self, v) -> "Expr":
        return self._cmp(v, CmpOp.Lte)

    def gt(self, v) -> "Expr":
        return self._cmp(v, CmpOp.Gt)

    def gte(self, v) -> "Expr":
        return self._cmp(v, CmpOp.Gte)

    def and_(self, other: "Expr") -> "Expr":
        return Expr("And", self, other)

    def or_(self, other: "Expr") -> "Expr":
        return Expr("Or", self, other)

    __and__ = and_
    __or__ = or_

    def __eq__(self, other):  # structural equality (the reference derives PartialEq)
        return (isinstance(other, Expr) and self.kind == other.kind and self.op == other.op and self.a == other.a
                and self.b == other.b)

    __hash__ = None

    def __repr__(self):
        return f"Expr({self.kind}, {self.a!r}, {self.b!r}, {self.op})"

    def compile(self, schema: Dict[str, DataType]) -> CompiledFilter:  # src/expr.rs:285-298
        return CompiledFilter(_normalize_plan(_lower_to_plan(self, schema)))


def col(name: str) -> Expr:  # src/expr.rs:108-111
    return Expr("Column", name)


def lit(v) -> Expr:  # src/expr.rs:112-115, Literal conversions :51-80
    if isinstance(v, Expr):
        return v
    if isinstance(v, bool):
        raise ExprError("Invalid literal")
    if isinstance(v, int):
        return Expr("Literal", ("I64", int(v)))
    if isinstance(v, float):
        return Expr("Literal", ("F64", float(v)))
    if isinstance(v, str):
        return Expr("Literal", ("Str", v))
    try:
        import numpy as np
        if isinstance(v, np.integer):
            return Expr("Literal", ("I64", int(v)))
        if isinstance(v, np.floating):
            return Expr("Literal", ("F64", float(v)))
    except ImportError:
        pass
    raise ExprError("Invalid literal")


def _normalize_plan(plan):  # src/expr.rs:300-343: drop `(c == v) OR (c != v)` tautologies
    out = []
    for clause in plan:
        taut = False
        for lf in clause:
            if lf.cmp == CmpOp.Eq and any(x.kind == lf.kind and x.cmp == CmpOp.Neq and x.column == lf.column and x.rhs == lf.rhs
                                          for x in clause):
                taut = True
                break
        if not taut:
            out.append(clause)
    return out


def _lower_to_plan(e: Expr, schema):  # src/expr.rs:355-372
    if e.kind == "And":
        a, b = _lower_to_plan(e.a, schema), _lower_to_plan(e.b, schema)
        if not a:
            return b
        if not b:
            return a
        return a + b  # and_concat_clauses, src/expr.rs:474-483
    if e.kind == "Or":
        a, b = _lower_to_plan(e.a, schema), _lower_to_plan(e.b, schema)
        if not a:
            return b
        if not b:
            return a
        return [ca + cb for ca in a for cb in b]  # or_distribute_clauses, src/expr.rs:494-511
    if e.kind == "Cmp":
        return [[_compile_cmp_leaf(e.a, e.b, e.op, schema)]]
    raise ExprError("Invalid expression (unexpected literal or column without comparator)")


def _compile_cmp_leaf(left: Expr, right: Expr, op: CmpOp, schema) -> ColumnFilter:  # src/expr.rs:385-466
    if not (left.kind == "Column" and right.kind == "Literal"):
        raise ExprError("Invalid expression shape for comparison (expect column vs literal)")
    name = left.a
    lk, lv = right.a
    if name not in schema:
        raise ExprError(f"Unknown column '{name}'")
    dt = schema[name]

    def mismatch(got):
        return ExprError(f"Type mismatch for column '{name}': expected {dt.name}, got literal {got}")

    if dt == DataType.String:
        if op not in (CmpOp.Eq, CmpOp.Neq):
            raise ExprError(f"Unsupported comparator for string column '{name}'")
        if lk != "Str":
            raise mismatch("string")
        return ColumnFilter("String", name, op, lv)
    if dt in (DataType.Int32, DataType.Int64):
        if lk == "F64":
            raise mismatch("float")
        if lk == "Str":
            raise mismatch("string")
        return ColumnFilter("Numeric", name, op, NumericLiteral("I64", lv))
    if dt == DataType.DateTime:
        if lk != "Str":
            raise mismatch("datetime string")
        ms = parse_datetime_millis(lv)
        if ms is None:
            raise mismatch("datetime string")
        return ColumnFilter("Numeric", name, op, NumericLiteral("I64", ms))
    # Float32 / Float64: ints are widened
    if lk == "Str":
        raise mismatch("string")
    return ColumnFilter("Numeric", name, op, NumericLiteral("F64", float(lv)))
