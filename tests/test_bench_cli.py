"""CPU: bench.py's launch contract.  `--gpus N` must never quietly run fewer ranks than asked for: a mismatch with the
launcher's WORLD_SIZE is an error, and on a box that cannot host N ranks the self-launched run fails loudly (non-zero
exit, no JSON line) instead of printing an n_gpus=1 line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_must_match_world_size():
    for gpus, world in (("2", "4"), ("8", "1"), ("1", "2")):
        r = subprocess.run([sys.executable, BENCH, "--gpus", gpus], env=_env(WORLD_SIZE=world), capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and r.stdout.strip() == ""


def test_self_launch_fails_loudly_without_enough_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box could really host two ranks")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rows", "2048", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0, r.stdout
    assert "2-GPU run failed" in r.stderr
    assert "n_gpus" not in r.stdout  # no benchmark line from a run that did not happen


def test_parity_gate_rejects_a_wrong_result(oracle):
    """bench.parity_rescore — the gate in front of the timed region — accepts the oracle's own top-k of regenerated rows
    and ends the run (SystemExit, non-zero) when a score bit, the order or a row is wrong."""
    import numpy as np
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    from otters_amd._native import HIT_DTYPE
    dim, seed, n, k = 48, 0x07735, 5000, 10
    rows = oracle.rand_rows(0, n, dim, seed)
    q = np.random.default_rng(3).uniform(-1, 1, dim).astype(np.float32)
    ref = oracle.vec_query(rows, q, oracle.METRIC_COSINE, oracle.TAKE_MAX, k, ties=oracle.TIES_CANONICAL)
    hits = np.zeros(k, dtype=HIT_DTYPE)
    hits["index"], hits["score"], hits["query"] = ref["index"], ref["score"], 0
    bench.parity_rescore(hits, q, dim, seed)  # the truth passes
    bad = hits.copy()
    bad["score"][3] = np.nextafter(bad["score"][3], np.float32(2.0))  # one ulp off
    with pytest.raises(SystemExit) as e:
        bench.parity_rescore(bad, q, dim, seed)
    assert "PARITY FAILED" in str(e.value)
    bad = hits.copy()
    bad[[2, 3]] = bad[[3, 2]]  # right rows and scores, wrong order
    with pytest.raises(SystemExit):
        bench.parity_rescore(bad, q, dim, seed)
    bad = hits.copy()
    bad["index"][0] = bad["index"][1]  # a row twice
    with pytest.raises(SystemExit):
        bench.parity_rescore(bad, q, dim, seed)
    bad = hits.copy()
    bad["index"][5] = (int(bad["index"][5]) + 1) % n  # a different row under the same score
    with pytest.raises(SystemExit):
        bench.parity_rescore(bad, q, dim, seed)
