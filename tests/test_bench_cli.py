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
