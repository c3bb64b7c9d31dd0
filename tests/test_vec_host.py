"""CPU: host logic of the VecStore mirror that needs no GPU — QueryBatch forms (src/vec.rs:320-336), plan validation and its
error strings (src/vec.rs:170-203), lowering to the resolved query handed to ott_query."""
import numpy as np
import pytest

from otters_amd import Cmp, Metric, Mode, OttersError, QueryBatch, TakeType, VecStore


def test_query_batch_forms():
    one = QueryBatch([1.0, 2.0, 3.0]).queries               # From<Vec<f32>>
    assert len(one) == 1 and one[0].dtype == np.float32 and one[0].tolist() == [1.0, 2.0, 3.0]
    many = QueryBatch([[1.0, 2.0], [3.0, 4.0]]).queries       # From<Vec<Vec<f32>>>
    assert len(many) == 2 and many[1].tolist() == [3.0, 4.0]
    mat = np.arange(12, dtype=np.float64).reshape(4, 3)
    kept = QueryBatch(mat).queries                            # a matrix stays one: a sequence of row vectors, f32, contiguous
    assert isinstance(kept, np.ndarray) and kept.dtype == np.float32 and kept.flags["C_CONTIGUOUS"]
    assert len(kept) == 4 and len(kept[2]) == 3 and kept[2].tolist() == [6.0, 7.0, 8.0]
    assert len(QueryBatch(np.ones(5, np.float32)).queries) == 1
    assert len(QueryBatch([]).queries) == 0


def test_resolve_matrix_and_list_agree():
    store = VecStore(3)
    store._n = 10  # host-side length only: resolve() never touches the GPU
    mat = np.random.default_rng(0).normal(size=(5, 3)).astype(np.float32)
    a = store.query(mat, Metric.Cosine).filter(0.25, Cmp.Gte).take_min(4).per_query().resolve()
    b = store.query([row.tolist() for row in mat], Metric.Cosine).filter(0.25, Cmp.Gte).take_min(4).per_query().resolve()
    assert np.array_equal(a.queries, b.queries) and a.queries.flags["C_CONTIGUOUS"]
    assert (a.metric, a.take, a.k, a.filter_cmp, a.filter_thr, a.mode) == (b.metric, b.take, b.k, b.filter_cmp, b.filter_thr, b.mode)
    assert a.take == int(TakeType.Min) and a.mode == int(Mode.PerQuery) and a.k == 4
    assert store.query(mat, Metric.Cosine).resolve().k == 10  # no take(): take_count = n_vecs (src/vec.rs:213)


def test_validation_errors_for_matrix_and_list():
    store = VecStore(3)
    store._n = 10
    with pytest.raises(OttersError, match="Query vector length 4 does not match expected dimension 3"):
        store.query(np.ones((2, 4), np.float32), Metric.Cosine).take(1).resolve()
    with pytest.raises(OttersError, match="Query vector length 2 does not match expected dimension 3"):
        store.query([[1.0, 2.0, 3.0], [1.0, 2.0]], Metric.Cosine).take(1).resolve()
    with pytest.raises(OttersError, match="No queries provided"):
        store.query([], Metric.Cosine).take(1).resolve()
    with pytest.raises(OttersError, match="No queries provided"):
        store.query(np.zeros((0, 3), np.float32), Metric.Cosine).take(1).resolve()
