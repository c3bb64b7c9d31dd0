"""Host-side mirror of otters' `col` module (src/col.rs): typed columns with a null mask.

Storage follows the reference exactly: a typed value vector plus a null bitmap (True = NULL),
with the reference's sentinels in the value slot of a NULL (i32::MIN / i64::MIN / NaN / "",
src/col.rs:238-326).  DateTime values are epoch milliseconds (UTC).
"""
from __future__ import annotations

import enum
import re
from datetime import datetime, timedelta, timezone
from typing import List, Optional

import numpy as np


class DataType(enum.IntEnum):  # src/type_utils.rs:11-19 (values = ott_dtype in the C ABI)
    Int32 = 0
    Int64 = 1
    Float32 = 2
    Float64 = 3
    String = 4
    DateTime = 5


class ColumnError(Exception):
    """src/col.rs:30-34, 86-97"""


_NP = {DataType.Int32: np.int32, DataType.Int64: np.int64, DataType.Float32: np.float32, DataType.Float64: np.float64,
       DataType.DateTime: np.int64}
_SENTINEL = {DataType.Int32: np.iinfo(np.int32).min, DataType.Int64: np.iinfo(np.int64).min, DataType.Float32: np.nan,
             DataType.Float64: np.nan, DataType.DateTime: np.iinfo(np.int64).min}

_EPOCH = datetime(1970, 1, 1, tzinfo=timezone.utc)
_RFC3339 = re.compile(r"^(\d{4})-(\d{2})-(\d{2})[Tt ](\d{2}):(\d{2}):(\d{2})(\.\d+)?([Zz]|[+-]\d{2}:\d{2})$")


def _millis(dt: datetime) -> int:
    return (dt - _EPOCH) // timedelta(milliseconds=1)


def parse_datetime_millis(s: str) -> Optional[int]:
    """src/col.rs:506-527 / src/expr.rs:267-283: RFC3339, then YYYY-MM-DD, then YYYY-MM-DD HH:MM:SS (UTC)."""
    m = _RFC3339.match(s)
    if m:
        try:
            y, mo, d, h, mi, sec = (int(m.group(i)) for i in range(1, 7))
            frac = m.group(7) or ""
            nanos = int((frac[1:] + "000000000")[:9]) if frac else 0
            off = m.group(8)
            if off in ("Z", "z"):
                tz = timezone.utc
            else:
                sign = 1 if off[0] == "+" else -1
                tz = timezone(sign * timedelta(hours=int(off[1:3]), minutes=int(off[4:6])))
            leap = sec == 60
            dt = datetime(y, mo, d, h, mi, 59 if leap else sec, tzinfo=tz)
            return _millis(dt.astimezone(timezone.utc)) + (1000 if leap else 0) + nanos // 1_000_000
        except ValueError:
            return None
    try:
        if re.fullmatch(r"\d{4}-\d{1,2}-\d{1,2}", s):
            return _millis(datetime.strptime(s, "%Y-%m-%d").replace(tzinfo=timezone.utc))
    except ValueError:
        pass
    try:
        return _millis(datetime.strptime(s, "%Y-%m-%d %H:%M:%S").replace(tzinfo=timezone.utc))
    except ValueError:
        return None


def _parse_datetime_fmt(s: str, fmt: str) -> int:  # src/col.rs:529-545
    try:
        return _millis(datetime.strptime(s, fmt).replace(tzinfo=timezone.utc))
    except ValueError:
        raise ColumnError(f"Parse error: Cannot parse '{s}' with format '{fmt}'")


class Column:
    """src/col.rs:21-28, 195-503"""

    def __init__(self, name: str, dtype: DataType):
        self._name = name
        self._dtype = DataType(dtype)
        self._vals: list = []
        self._nulls: List[bool] = []
        self._np: Optional[np.ndarray] = None
        self._np_nulls: Optional[np.ndarray] = None
        self._datetime_format: Optional[str] = None

    @staticmethod
    def new(name: str, dtype: DataType) -> "Column":
        return Column(name, dtype)

    @staticmethod
    def from_numpy(name: str, dtype: DataType, values: np.ndarray, nulls: Optional[np.ndarray] = None) -> "Column":
        """Bulk constructor (extension): values already in storage form, nulls = bool array (True = NULL)."""
        c = Column(name, dtype)
        if DataType(dtype) == DataType.String:
            c._vals = [str(v) for v in values]
            c._nulls = [bool(x) for x in nulls] if nulls is not None else [False] * len(values)
            return c
        c._np = np.ascontiguousarray(values, dtype=_NP[DataType(dtype)]).copy()
        c._np_nulls = np.zeros(c._np.size, bool) if nulls is None else np.ascontiguousarray(nulls, dtype=bool).copy()
        if nulls is not None:
            c._np[c._np_nulls] = _SENTINEL[DataType(dtype)]
        return c

    def name(self) -> str:
        return self._name

    def dtype(self) -> DataType:
        return self._dtype

    def len(self) -> int:
        return len(self._vals) + (self._np.size if self._np is not None else 0)

    def __len__(self) -> int:
        return self.len()

    def is_empty(self) -> bool:
        return self.len() == 0

    def with_datetime_fmt(self, fmt: str) -> "Column":  # src/col.rs:352-355
        self._datetime_format = fmt
        return self

    def _flush(self) -> None:
        """merge list-pushed values into the numpy storage"""
        if self._dtype == DataType.String or not self._vals:
            return
        add = np.array(self._vals, dtype=_NP[self._dtype])
        addn = np.array(self._nulls, dtype=bool)
        if self._np is None:
            self._np, self._np_nulls = add, addn
        else:
            self._np = np.concatenate([self._np, add])
            self._np_nulls = np.concatenate([self._np_nulls, addn])
        self._vals, self._nulls = [], []

    def push(self, value) -> None:
        """Unified push (src/col.rs:357-390).  None = NULL.  DateTime columns take epoch millis or
        a datetime string (parsed with the column's format if one was set)."""
        dt = self._dtype
        if value is None:
            self._vals.append("" if dt == DataType.String else _SENTINEL[dt])
            self._nulls.append(True)
            return
        if dt == DataType.String:
            if not isinstance(value, str):
                raise ColumnError(f"Type mismatch: expected {dt.name}, got incompatible type")
            self._vals.append(value)
        elif dt == DataType.DateTime:
            if isinstance(value, str):
                if self._datetime_format:
                    ms = _parse_datetime_fmt(value, self._datetime_format)
                else:
                    ms = parse_datetime_millis(value)
                    if ms is None:
                        raise ColumnError(f"Parse error: Cannot parse '{value}' as datetime. Supported formats: ISO 8601, "
                                          "YYYY-MM-DD, YYYY-MM-DD HH:MM:SS")
                self._vals.append(ms)
            elif isinstance(value, (int, np.integer)) and not isinstance(value, bool):
                self._vals.append(int(value))
            else:
                raise ColumnError(f"Type mismatch: expected {dt.name}, got incompatible type")
        elif dt in (DataType.Int32, DataType.Int64):
            if isinstance(value, bool) or not isinstance(value, (int, np.integer)):
                raise ColumnError(f"Type mismatch: expected {dt.name}, got incompatible type")
            self._vals.append(int(value))
        else:
            if isinstance(value, bool) or not isinstance(value, (int, float, np.integer, np.floating)):
                raise ColumnError(f"Type mismatch: expected {dt.name}, got incompatible type")
            self._vals.append(float(value))
        self._nulls.append(False)

    def from_(self, values) -> "Column":  # Column::from, src/col.rs:392-401 (`from` is a Python keyword)
        for v in values:
            self.push(v)
        return self

    # -- typed accessors (src/col.rs:446-502) ----------------------------------------------------
    def values(self):
        if self._dtype == DataType.String:
            return self._vals
        self._flush()
        if self._np is None:
            return np.zeros(0, dtype=_NP[self._dtype])
        return self._np

    def null_mask(self) -> np.ndarray:
        """bool array, True = NULL (BitVec<usize,Lsb0> in the reference)"""
        if self._dtype == DataType.String:
            return np.array(self._nulls, dtype=bool)
        self._flush()
        if self._np_nulls is None:
            return np.zeros(0, dtype=bool)
        return self._np_nulls

    def _typed(self, dt: DataType):
        return self.values() if self._dtype == dt else None

    def i32_values(self):
        return self._typed(DataType.Int32)

    def i64_values(self):
        return self._typed(DataType.Int64)

    def f32_values(self):
        return self._typed(DataType.Float32)

    def f64_values(self):
        return self._typed(DataType.Float64)

    def string_values(self):
        return self._typed(DataType.String)

    def datetime_values(self):
        return self._typed(DataType.DateTime)

    def get(self, i: int):
        """value at row i (None for NULL) — convenience for tests and display"""
        if self._nulls[i] if self._dtype == DataType.String else self.null_mask()[i]:
            return None
        v = self.values()[i]
        return v if self._dtype == DataType.String else v.item()

    def take(self, indices) -> "Column":
        """gather rows into a new column (result materialisation, src/meta.rs:728-821)"""
        idx = np.asarray(indices, dtype=np.int64)
        if self._dtype == DataType.String:
            # only the taken rows are looked at: building the whole column's null mask here cost 22 ms per query on a 2M-row store
            return Column.from_numpy(self._name, self._dtype, [self._vals[i] for i in idx], np.array([self._nulls[i] for i in idx], dtype=bool))
        return Column.from_numpy(self._name, self._dtype, self.values()[idx], self.null_mask()[idx])

    def data_type(self) -> DataType:  # ColumnValues::data_type, src/col.rs:74-83 (what `col.values().data_type()` answers)
        return self._dtype

    def head_n(self, n: int) -> str:  # src/col.rs:409-443
        return self.head(n)

    def head(self, n: int = 5) -> str:  # src/col.rs:403-444
        lines = [f"Column: {self._name} ({self._dtype.name})"]
        for i in range(min(self.len(), n)):
            v = self.get(i)
            if v is None:
                lines.append(f"  [{i}]: NULL")
            elif self._dtype in (DataType.Float32, DataType.Float64):
                lines.append(f"  [{i}]: {v:.4f}")
            elif self._dtype == DataType.String:
                lines.append(f"  [{i}]: \"{v}\"")
            elif self._dtype == DataType.DateTime:
                lines.append(f"  [{i}]: {format_datetime(v)} ({v})")
            else:
                lines.append(f"  [{i}]: {v}")
        if self.len() > n:
            lines.append(f"  ... ({self.len() - n} more rows)")
        out = "\n".join(lines)
        print(out)
        return out


def format_datetime(ms: int) -> str:
    try:
        return (_EPOCH + timedelta(milliseconds=int(ms))).strftime("%Y-%m-%d %H:%M:%S UTC")
    except (OverflowError, ValueError):
        return f"Invalid timestamp"
