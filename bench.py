#!/usr/bin/env python3
"""bench.py — headline benchmark of the otters hot path on MI355X.

Workload (BASELINE.json `metric`): exact cosine top-10 over a 10M x 768 f32 corpus resident in
HBM, one query per step (`VecStore.query(q, Metric::Cosine).take(10).collect()`), synthetic
uniform [-1,1) rows (the distribution of examples/demo.rs).  With --gpus N each rank owns a
10M-row shard of an N*10M-row corpus (weak scaling); every step scores the query on every
shard and all-gathers the per-GPU top-k over RCCL for the final merge.

One JSON line on rank 0; `value` = GB/s scanned by the whole job (queries/sec beside it).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured float4-copy ceiling


def cpu_baseline(dim: int, k: int, seed: int) -> dict:
    """The oracle (a restatement of the reference's single-threaded VecQueryPlan::collect
    loop, src/vec.rs:223-267) timed on one host core over a bounded sample of the workload."""
    import oracle as O  # the one place bench.py touches the oracle: a reported baseline, never the product
    n = 1_000_000 if dim <= 768 else 200_000
    rows = O.rand_rows(0, n, dim, seed)
    inv = O.inv_norms(rows)
    q = np.random.default_rng(seed + 1).uniform(-1, 1, (1, dim)).astype(np.float32)
    O.vec_query(rows[:1000], q, O.METRIC_COSINE, O.TAKE_MAX, k, inv=inv[:1000], fast=True)
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or time.perf_counter() - t0 < 10.0:
        O.vec_query(rows, q, O.METRIC_COSINE, O.TAKE_MAX, k, inv=inv, fast=True)
        reps += 1
        if reps >= 400:  # (bounded either way: ~10 s of one core)
            break
    dt = (time.perf_counter() - t0) / reps
    gb = n * (dim * 4 + 4) / 1e9
    return {"value": round(gb / dt, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{n}x{dim} f32 rows (1/{10_000_000 // n} of the workload), single-query cosine top-{k}, "
                      f"{reps} reps, oracle C port built -O3 -mavx2, 1 thread as src/vec.rs:223",
            "queries_per_sec_at_sample": round(1.0 / dt, 3),
            "queries_per_sec_extrapolated_10M": round(1.0 / dt * n / 10_000_000, 4)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0x07735)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational config-2 batch measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    from otters_amd import Metric, VecStore
    from otters_amd.dist import ShardedVecStore

    # OTT_BENCH_SINGLE_DEVICE=1: every rank uses GPU 0 and gloo carries the candidate blocks — a functional check of the
    # N-rank path on a 1-GPU box (RCCL refuses two ranks on one device); never a performance configuration
    single_dev = os.environ.get("OTT_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
        os.environ.setdefault("OTT_BENCH_BACKEND", "gloo")
    dist = None
    # OTT_BENCH_FORCE_DIST=1 runs the sharded code path (ott_query_device -> RCCL all-gather -> merge kernel)
    # even with one rank, so it can be exercised on a 1-GPU box
    if world > 1 or os.environ.get("OTT_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("OTT_BENCH_BACKEND", "nccl")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                probe = torch.zeros(1, device=f"cuda:{local_rank}")
                dist.all_reduce(probe)  # fail here, not in the timed loop, if RCCL cannot talk
                torch.cuda.synchronize(local_rank)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:  # keep the scaling run alive: candidate blocks are k*16 bytes, gloo can carry them
            print(f"[bench] RCCL unavailable ({e!r}); exchanging candidates over gloo", file=sys.stderr, flush=True)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo", rank=rank, world_size=world)

    store = VecStore(args.dim, device=local_rank)
    store.set_base_offset(rank * args.rows)
    store.reserve(args.rows)
    store.append_random(args.rows, args.seed)
    sharded = ShardedVecStore(store, dist) if dist is not None else None

    rng = np.random.default_rng(args.seed + 1)
    queries = rng.uniform(-1, 1, (args.steps + args.warmup, args.dim)).astype(np.float32)

    def step(i: int):
        q = queries[i]
        if sharded is not None:
            return sharded.query(q, Metric.Cosine).take(args.k).collect()
        return store.query(q, Metric.Cosine).take(args.k).collect()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(local_rank)
        store_sync(store)

    for i in range(args.warmup):
        step(i)
    barrier()
    kernel_ns = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(args.warmup + i)
        kernel_ns.append(store.last_stats["score_ns"])  # hipEvent time of the scoring kernel on the store's stream
    barrier()
    dt = time.perf_counter() - t0
    assert len(res) == args.k

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        bytes_per_pass = args.rows * (args.dim * 4 + 4)  # algorithmic: 4*dim per row + 4 B inverse norm (cosine)
        ms_per_step = dt / args.steps * 1e3
        qps = args.steps / dt
        gbs = world * bytes_per_pass * qps / 1e9
        kern_ms = float(np.mean(kernel_ns)) / 1e6
        achieved = bytes_per_pass / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "GB/s scanned + queries/sec, exact cosine top-10 over 10M x 768 f32 rows per GPU",
            "value": round(gbs, 2), "unit": "GB/s", "queries_per_sec": round(qps, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.rows}x{args.dim} f32 VecStore per GPU, single query, Metric::Cosine, take({args.k})",
                       "rows_per_gpu": args.rows, "dim": args.dim, "k": args.k, "nq": 1,
                       "sharding": "none" if world == 1 else f"{world} row shards, {'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()} all-gather of per-GPU top-{args.k}",
                       "path": "exact-order VALU scorer + fused wavefront top-k"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE x2 gfx950
                         # correction + WRITE_SIZE; profiles/round1/bench_n1_pmc_*.csv); only valid for the default workload
                         "traffic": 30.76e9 if (args.rows, args.dim) == (10_000_000, 768) else None,
                         "kernel": "ott::exact_kernel<false, 1, 1, false, false, false>", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": bytes_per_pass},
        }
        if world == 1 and not args.no_extras:
            # outside the timed region, informational: BASELINE config 2 (256 queries, top-100, merged) on the same resident
            # corpus through the batch path (certified cascade: bf16 hi-plane candidate pass on the matrix cores, split-bf16
            # pass for what it cannot certify, exact f32 re-score of every candidate)
            try:
                Q = rng.uniform(-1, 1, (256, args.dim)).astype(np.float32)
                store.query(Q, Metric.Cosine).take(100).collect_arrays()  # builds the batch image on first use
                t1 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    store.query(Q, Metric.Cosine).take(100).collect_arrays()
                bdt = (time.perf_counter() - t1) / reps
                line["extras"] = {"config2_256q_top100_ms_per_batch": round(bdt * 1e3, 3),
                                  "config2_queries_per_sec": round(256 / bdt, 1),
                                  "config2_score_phase_ms": round(store.last_stats["score_ns"] / 1e6, 3),
                                  "config2_f32_equiv_tflops": round(2.0 * args.rows * args.dim * 256 / (store.last_stats["score_ns"] * 1e-9) / 1e12, 1),
                                  "config2_queries_refined_split_pass": int(store.last_stats["refined"]),
                                  "config2_queries_rerun_exact": int(store.last_stats["retries"])}
                # opt-in, also informational: ONE query through the same cascade (Path.Mfma) instead of the exact-order kernel
                # the timed region above measures (AUTO keeps single queries on that kernel: no second copy of the corpus)
                from otters_amd import Path
                q1 = queries[0]
                store.query(q1, Metric.Cosine).take(args.k).with_path(Path.Mfma).collect()
                t1 = time.perf_counter()
                for i in range(10):
                    store.query(queries[i % len(queries)], Metric.Cosine).take(args.k).with_path(Path.Mfma).collect()
                line["extras"]["single_query_via_cascade_ms"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
            except Exception as e:  # noqa: BLE001 -- never let the informational part break the contract line
                line["extras"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.dim, args.k, args.seed)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def store_sync(store) -> None:
    from otters_amd import _native as N
    if store._h is not None:
        N.check(N.lib().ott_store_sync(store._h))


if __name__ == "__main__":
    main()
