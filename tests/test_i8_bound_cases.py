"""CPU: the adversarial corpora of tests/adversarial_i8.py do what they claim — the emulation they are built with agrees with the
oracle bit for bit, the oracle (= the reference's arithmetic, src/vec_compute.rs:9-22) returns the drifting A rows, and a candidate
pass certified with round 5's bound of the int8 level (16 * 2^-24 + the measured quantisation losses: the approximate side only)
would have returned the P rows as certified.  The GPU half is tests/test_gpu_i8_bound.py."""
import numpy as np
import pytest

import adversarial_i8 as A


@pytest.mark.parametrize("dim,metric,signed", [(768, "cosine", False), (768, "dot", True), (1030, "cosine", True), (1030, "dot", False)])
def test_case_is_adversarial_for_the_old_bound(oracle, dim, metric, signed):
    case = A.build_case(dim, metric, seed=11, n=6000, signed=signed)
    rows, q, info = case["rows"], case["query"], case["info"]
    k = info["k"]
    m = oracle.METRIC_COSINE if metric == "cosine" else oracle.METRIC_DOT
    # the emulation is the oracle's arithmetic: same bits on a sample of rows (drifting, quiet, filler)
    sample = np.r_[0:8, info["copies"]:info["copies"] + 12, 3000:3040, rows.shape[0] - 10:rows.shape[0]]
    emu, _ = A.exact_scores(rows[sample], q, metric)
    inv_q = oracle.inv_norms(q[None, :])[0]
    inv_v = oracle.inv_norms(rows[sample])
    for j, r in enumerate(sample):
        ref = oracle.cosine(q, rows[r], inv_q, inv_v[j]) if metric == "cosine" else oracle.dot(q, rows[r])
        assert np.float32(ref).view(np.uint32) == np.float32(emu[j]).view(np.uint32), (r, ref, emu[j])
    # every row and the query are int8-representable: the measured losses are rounding noise
    assert info["i8_rel"] < 2e-7 and info["qrel"] < 2e-7, info
    assert min(info["margins_units"]) > 3.0, info
    ref = oracle.vec_query(rows, q[None, :], m, oracle.TAKE_MAX, k, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(ref["index"], case["expect_rows"])
    for T in (128, 512):  # the sweep's / the tile's list at k = 10, and the list once a store's queries have failed at that
        top, certified = A.old_bound_outcome(rows, q, metric, k, T, info["eps_old"])
        assert certified and not np.array_equal(np.sort(top), ref["index"].astype(np.int64)), (T, top, ref["index"])
    # the drift the construction leans on: A's exact-order score exceeds its integer-exact approximate score by well over the old bound
    assert info["exact_A"] - info["approx_A"] > 1.5 * info["eps_old"], info
