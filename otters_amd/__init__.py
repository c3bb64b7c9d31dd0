"""otters_amd — MI355X (gfx950) backend for the otters exact-vector-search hot path.

Host-side mirror of the reference's `vec` / `meta` modules (same names, argument meaning
and error strings) over libotters_hip.so: the corpus lives row-major in HBM, scoring,
score filtering and top-k run as hand-written HIP kernels.  No CPU fallback.
"""
from ._native import OttersError  # noqa: F401
from .vec import (Cmp, Metric, Mode, Path, QueryBatch, SearchResult, TakeType, VecQueryPlan,  # noqa: F401
                  VecStore)

from .col import Column, ColumnError, DataType  # noqa: F401,E402
from .expr import CmpOp, CompiledFilter, Expr, ExprError, col, lit  # noqa: F401,E402
from .meta import (MetaBuildStats, MetaQueryPlan, MetaQueryResults, MetaQueryStats, MetaStore,  # noqa: F401,E402
                   MetaStoreBuilder)

__all__ = ["Column", "ColumnError", "DataType", "CmpOp", "CompiledFilter", "Expr", "ExprError", "col", "lit",
           "MetaBuildStats", "MetaQueryPlan", "MetaQueryResults", "MetaQueryStats", "MetaStore", "MetaStoreBuilder",
           "OttersError", "Cmp", "Metric", "Mode", "Path", "QueryBatch", "SearchResult", "TakeType",
           "VecQueryPlan", "VecStore"]
