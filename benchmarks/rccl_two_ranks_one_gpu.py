"""Probe: RCCL with world > 1 on a box that has ONE GPU.

RCCL refuses two ranks of one communicator on the same device of the same HOST ("Duplicate GPU detected"); the host is
identified by a hash that NCCL_HOSTID overrides.  With a different NCCL_HOSTID per rank the two processes look like two
nodes with one GPU each: ncclCommInitRank(world = 2) runs its real bootstrap, and ncclAllGather carries the candidate
blocks through RCCL's network transport (TCP over `lo`) between two RCCL kernels — everything of the N > 1 path except xGMI.

    python benchmarks/rccl_two_ranks_one_gpu.py [world] [rows_per_rank] [dim]

Parent: starts `world` children (never touches the GPU itself), prints what they report.  Child r: shard r of a seeded
corpus in HBM (device 0), ott_query_sharded over an RCCL comm, results compared with a one-store query of the whole corpus
(same process, no comm): indices, owners and f32 score bits must be equal."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(rank, world, n_per, dim, uid_path):
    sys.path.insert(0, ROOT)
    import numpy as np
    from otters_amd import Metric, VecStore
    from otters_amd.dist import Comm, ShardedVecStore
    if rank == 0:
        uid = Comm.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 60:
                raise SystemExit("rank 0 never published the RCCL id")
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    t0 = time.time()
    comm = Comm.rccl(uid, rank, world, 0)
    t_init = time.time() - t0
    rng = np.random.default_rng(4242)
    rows = rng.uniform(-1, 1, (world * n_per, dim)).astype(np.float32)
    rows[::977] = rows[5]  # exact score ties across shards
    queries = rng.uniform(-1, 1, (9, dim)).astype(np.float32)
    shard = VecStore(dim)
    shard.add_vectors(rows[rank * n_per:(rank + 1) * n_per])
    shard.set_base_offset(rank * n_per)
    sh = ShardedVecStore(shard, comm)
    whole = VecStore(dim)
    whole.add_vectors(rows)
    report = dict(rank=rank, world=world, transport=comm.transport, init_s=round(t_init, 2), global_rows=sh.len(), cases=[])
    for metric, k, perq, nq in ((Metric.Cosine, 10, False, 1), (Metric.DotProduct, 100, False, 9), (Metric.Euclidean, 64, True, 9),
                                (Metric.Cosine, 300, True, 9), (Metric.Cosine, 700, False, 9)):
        plan_s, plan_w = sh.query(queries[:nq], metric).take(k), whole.query(queries[:nq], metric).take(k)
        if perq:
            plan_s, plan_w = plan_s.per_query(), plan_w.per_query()
        t0 = time.time()
        got, got_counts = plan_s.collect_arrays()
        dt = time.time() - t0
        want, want_counts = plan_w.collect_arrays()
        ok = (got.shape == want.shape and np.array_equal(got["index"], want["index"]) and np.array_equal(got["query"], want["query"])
              and np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32)) and (not perq or [int(c) for c in got_counts] == [int(c) for c in want_counts]))
        report["cases"].append(dict(metric=metric.name, k=k, per_query=perq, nq=nq, hits=int(got.size), equal=bool(ok), ms=round(dt * 1e3, 2)))
    # the raw collective
    g = comm.all_gather_host(np.array([rank + 1, 1000 + rank], dtype=np.int64))
    report["all_gather_host"] = g.reshape(-1).tolist()
    print("REPORT " + json.dumps(report), flush=True)
    sh_ok = all(c["equal"] for c in report["cases"]) and report["all_gather_host"] == [v for r in range(world) for v in (r + 1, 1000 + r)]
    comm.close()
    os._exit(0 if sh_ok else 1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
        return
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    tmp = tempfile.mkdtemp(prefix="ott_rccl_")
    uid_path = os.path.join(tmp, "uid")
    procs = []
    for r in range(world):
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        env.update(NCCL_HOSTID=f"ott-probe-node-{r}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"),
                   OTT_COMM_TIMEOUT_MS="60000", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(world), str(n_per), str(dim), uid_path],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    rc, deadline = 0, time.time() + 240
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=max(deadline - time.time(), 1))
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[probe] TIMEOUT: killed"
        print(f"--- rank {r} (exit {p.returncode}) ---\n{out[-6000:]}")
        rc |= 1 if p.returncode else 0
    print("PROBE", "OK" if rc == 0 else "FAILED")
    sys.exit(rc)


if __name__ == "__main__":
    main()
