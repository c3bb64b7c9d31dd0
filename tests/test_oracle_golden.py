"""CPU: pins the oracle (oracle/otters_oracle.c) against every known-answer case the
reference's own tests hold for the hot path (tests/golden/, transcribed from
tests/vec_store_tests.rs), driving it through the host layer's plan logic."""
import numpy as np
import pytest

from helpers import check_expect, load, oracle_collect, plan_from_case
from otters_amd import OttersError, VecQueryPlan, VecStore

VEC_CASES = load("vec_store_cases.json")


class HostOnlyStore(VecStore):
    """VecStore whose rows stay in a numpy array: lets the CPU suite exercise the host plan
    logic (validation, take/filter resolution) and hand the resolved query to the ORACLE.
    Lives in tests/ only; the product VecStore always talks to libotters_hip.so."""

    def __init__(self, dim):
        super().__init__(dim)
        self.host_rows = np.zeros((0, dim), np.float32)

    def _append(self, rows):
        self.host_rows = np.concatenate([self.host_rows, np.asarray(rows, np.float32)])
        self._n = self.host_rows.shape[0]


@pytest.mark.parametrize("case", [c for c in VEC_CASES if "kernel" in c], ids=lambda c: c["name"])
def test_kernel_known_answers(oracle, case):
    for mode in (oracle.REDUCE_AVX, oracle.REDUCE_SEQ4):
        if case["kernel"] == "dot":
            got = oracle.dot(case["a"], case["b"], mode)
        elif case["kernel"] == "l2sq":
            got = oracle.l2sq(case["a"], case["b"], mode)
        else:
            got = oracle.cosine(case["a"], case["b"], case["inv_a"], case["inv_b"], mode)
        if case.get("exact"):
            assert got == np.float32(case["expect"])
        else:
            assert abs(float(got) - case["expect"]) < case["tol"]


@pytest.mark.parametrize("case", [c for c in VEC_CASES if "metric" in c], ids=lambda c: c["name"])
@pytest.mark.parametrize("ties", [0, 1], ids=["literal", "canonical"])
def test_vec_store_cases(oracle, case, ties):
    store = HostOnlyStore(case["dim"])
    exp = case["expect"]
    if case["vectors"]:
        store.add_vectors(case["vectors"])
    plan = plan_from_case(case, store)
    if "error_contains" in exp or "error_eq" in exp:
        with pytest.raises(OttersError) as ei:
            plan.collect()
        if "error_eq" in exp:
            assert str(ei.value) == exp["error_eq"]
        else:
            assert exp["error_contains"] in str(ei.value)
        return
    rq = plan.resolve()
    hits = oracle_collect(oracle, rq, store.host_rows, ties)
    check_expect(hits["index"], hits["score"], exp)


def test_plan_new_unset():
    case = next(c for c in VEC_CASES if c.get("plan_new"))
    with pytest.raises(OttersError) as ei:
        VecQueryPlan.new().collect()
    assert case["expect"]["error_contains"] in str(ei.value)
    # vec_store_tests.rs:1000-1019: filter / take* on an unset plan keep failing at collect
    for plan in (VecQueryPlan.new().filter(0.5, 2), VecQueryPlan.new().take(5), VecQueryPlan.new().take_min(5),
                 VecQueryPlan.new().take_max(5)):
        with pytest.raises(OttersError):
            plan.collect()


def test_add_vectors_dim_mismatch():
    case = next(c for c in VEC_CASES if c.get("add_vectors"))
    store = HostOnlyStore(case["add_vectors"]["dim"])
    with pytest.raises(OttersError) as ei:
        store.add_vectors(case["add_vectors"]["vectors"])
    assert case["expect"]["error_contains"] in str(ei.value)
    assert store.len() == 1  # try_for_each: the good row before the bad one stays (src/vec.rs:373-376)
    s2 = VecStore(3)
    with pytest.raises(OttersError):
        s2.add_vector([1.0, 2.0])  # vec_store_tests.rs:20-27
