#!/usr/bin/env python3
"""bench.py — headline benchmark of the otters hot path on MI355X.

Workload (BASELINE.json `metric`): exact cosine top-10 over a 10M x 768 f32 corpus resident in
HBM, one query per step (`VecStore.query(q, Metric::Cosine).take(10).collect()`), synthetic
uniform [-1,1) rows (the distribution of examples/demo.rs).  With --gpus N each rank owns a
10M-row shard of an N*10M-row corpus (weak scaling); every step scores the query on every
shard, all-gathers the per-GPU top-k over RCCL and merges (`ott_query_sharded`: one C-ABI call,
score -> ncclAllGather -> merge on one HIP stream).

Launching: `python bench.py --gpus N` starts its N ranks itself (fresh child processes, one per
GPU; the parent never touches a GPU) unless it already runs under a launcher (WORLD_SIZE set:
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`).  RCCL is mandatory
for N > 1: if the communicator cannot be created the run fails, it never falls back.

Before anything is timed the GPU result is checked against the CPU oracle (parity gate); a
mismatch ends the run with a non-zero exit code.  One JSON line on rank 0; `value` = GB/s
scanned by the whole job (queries/sec beside it).
"""
import argparse
import csv
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured float4-copy ceiling
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
I8_PEAK_TOPS = 5000.0      # dense int8 MFMA peak (MI355X_MICROARCH.md: the fp8 / int8 figure, twice the 16-bit one)
F32_MFMA_PEAK_TFLOPS = 157.3
SAMPLE_ROWS = 977 * 1024   # the CPU sample: whole chunks (default chunk size 1024), ~1/10 of the workload


# ------------------------------------------------------------------------------------------------
# launcher: N fresh children, one per GPU.  Runs before anything imports torch or touches HIP.
# ------------------------------------------------------------------------------------------------
def launch_ranks(args) -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OTT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout (the JSON line) is collected by a thread while every child is watched: the first rank that dies takes
    # the whole run down (its peers would otherwise wait for it in a rendezvous or a collective until some timeout)
    import signal
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop_children():
        for p in procs:  # exactly the processes started above, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    def on_term(signum, _frame):  # a driver timeout / Ctrl-C must not leave N ranks behind, each holding a GPU and a corpus
        raise KeyboardInterrupt(f"signal {signum}")
    old_handlers = {sig: signal.signal(sig, on_term) for sig in (signal.SIGTERM, signal.SIGINT)}
    failed = False
    try:
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                failed = True
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.05)
    except BaseException:
        stop_children()  # the parent never touched a GPU: nothing else to clean up
        raise
    finally:
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    if failed:
        stop_children()
    reader.join(timeout=10)
    codes = [p.returncode for p in procs]
    if failed:
        print(f"[bench] rank exit codes {codes}: the {args.gpus}-GPU run failed", file=sys.stderr)
        return 1
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------------------------------
# CPU side: the oracle as checker (parity gate) and as the reported baseline.  Nothing here is the product.
# ------------------------------------------------------------------------------------------------
def cpu_sample(dim: int, seed: int):
    import oracle as O
    n = SAMPLE_ROWS if dim <= 768 else 200 * 1024
    rows = O.rand_rows(0, n, dim, seed)
    return rows, O.inv_norms(rows)


def cpu_baseline(rows, inv, q, k: int) -> dict:
    """The oracle (a restatement of the reference's single-threaded VecQueryPlan::collect
    loop, src/vec.rs:223-267) timed on one host core over a bounded sample of the workload."""
    import oracle as O
    n, dim = rows.shape
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
    return {"value": round(gb / dt, 3), "unit": "GB/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{n}x{dim} f32 rows (~1/{round(10_000_000 / n)} of the workload), single-query cosine top-{k}, "
                      f"{reps} reps, oracle C port built -O3 -mavx2, 1 thread as src/vec.rs:223",
            "queries_per_sec_at_sample": round(1.0 / dt, 3),
            "queries_per_sec_extrapolated_10M": round(1.0 / dt * n / 10_000_000, 4)}


def host_ram_available() -> int:
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return 0


def cpu_baseline_all_cores(store, sample, q, k: int) -> dict:
    """Config 3's CPU side on EVERY host core: the oracle's restatement of MetaQueryPlan::collect's score + merge block (one task
    per surviving chunk on a thread pool, as rayon's par_iter does at src/meta.rs:678): chunk_size 4096, vec_filter(0.5, Gt),
    take(10).  The workload has 1221 surviving chunks; a sample with fewer tasks than threads starves the pool (round 4 timed 122
    chunks on 256 threads), so the sample holds AT LEAST FOUR surviving chunks per thread, host memory permitting — the chunks
    the zonemap prunes are never touched by the CPU path, so only survivors are materialised: the first rows of the store, read
    back from HBM (the same counter-based rows the parity sample regenerates on the host), every chunk of them a survivor.
    When the box has too little memory for that the 1M-row parity sample is used and `tasks_per_thread` says what it was."""
    import oracle as O
    rows, inv = sample
    dim = rows.shape[1]
    cs = 4096
    cores = os.cpu_count() or 1
    want_chunks = 4 * cores
    avail = host_ram_available()
    big = want_chunks * cs
    got_from = "the 1M-row parity sample"
    if store is not None and big > rows.shape[0] and store.len() >= big and avail > 2.2 * big * (dim * 4 + 4):
        rows = store.rows(0, big)          # (D2H of rows the GPU generated: bit-identical to the host generator's)
        inv = store.inv_norms(0, big)
        got_from = "the store's first rows, read back from HBM"
    n = rows.shape[0]
    n_chunks = (n + cs - 1) // cs
    # every second chunk pruned (config 3's zonemap) only when the sample is the small one; the large one is survivors only
    cmask = ((np.arange(n_chunks) % 2) == 1) if n < big else np.ones(n_chunks, dtype=bool)
    surviving = int(cmask.sum())
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or time.perf_counter() - t0 < 5.0:
        _, st = O.meta_query(rows, cs, q, O.METRIC_COSINE, O.TAKE_MAX, k, O.CMP_GT, 0.5, chunk_mask=cmask, n_threads=cores, inv=inv, fast=True)
        reps += 1
        if reps >= 2000:
            break
    dt = (time.perf_counter() - t0) / reps
    scored = int(st["vectors_compared"])
    gb = scored * (dim * 4 + 4) / 1e9
    return {"value": round(gb / dt, 3), "unit": "GB/s", "cores": cores, "host_cores": cores, "kind": "port",
            "tasks": surviving, "tasks_per_thread": round(surviving / cores, 2),
            "sample": f"{surviving} surviving chunks of {cs} rows x {dim} f32 ({scored} rows scored; {got_from}), single-query cosine, "
                      f"vec_filter(0.5, Gt), take({k}), {reps} reps, one task per surviving chunk on {cores} threads = "
                      f"{surviving / cores:.2f} tasks per thread (src/meta.rs:678; the full workload has 1221 surviving chunks)",
            "queries_per_sec_at_sample": round(1.0 / dt, 3),
            "queries_per_sec_extrapolated_config3": round(1.0 / dt * surviving / 1221.0, 3)}


def bits(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def parity_rescore(hits, q, dim: int, seed: int) -> None:
    """Every returned score re-derived by the oracle from the regenerated row (the corpus generator is counter-based:
    any global row can be rebuilt on the host), compared bit for bit; order and distinctness checked."""
    import oracle as O
    idx = hits["index"].astype(np.int64)
    if len(set(idx.tolist())) != idx.size:
        raise SystemExit("[bench] PARITY FAILED: duplicate rows in the result")
    rows = np.concatenate([O.rand_rows(int(i), 1, dim, seed) for i in idx]) if idx.size else np.zeros((0, dim), np.float32)
    ref = O.vec_query(rows, q, O.METRIC_COSINE, O.TAKE_MAX, idx.size, ties=O.TIES_CANONICAL)
    # ref is sorted best-first over the SAME rows; its order must be the returned order, its scores the returned bits
    got_order = [int(i) for i in idx]
    ref_order = [int(idx[j]) for j in ref["index"]]
    if got_order != ref_order or not np.array_equal(bits(hits["score"]), bits(ref["score"])):
        raise SystemExit(f"[bench] PARITY FAILED: returned scores differ from the oracle's re-score\n got {hits}\n ref {ref}")


def profile_counter(name: str, kernel_substr: str):
    """Mean per-dispatch value of a rocprofv3 PMC counter for the dispatches of one kernel, from the newest committed
    profile of this same command (profiles/roundN/bench_n1_pmc_<name>.csv).  Returns (value, relative path) or (None, None)."""
    pdir = os.path.join(ROOT, "profiles")
    rounds = sorted((d for d in os.listdir(pdir) if d.startswith("round")), reverse=True) if os.path.isdir(pdir) else []
    for rd in rounds:
        path = os.path.join(pdir, rd, f"bench_n1_pmc_{name}.csv")
        if not os.path.exists(path):
            continue
        vals = []
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == name:
                    vals.append(float(row["Counter_Value"]))
        if vals:
            return float(np.mean(vals)), os.path.relpath(path, ROOT)
    return None, None


def measure_traffic_live(args):
    """HBM bytes per launch of the headline kernel, measured for THIS run: two short child runs of this same script under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes: they do not fit the TCC counter slots together), started
    BEFORE this process touches a GPU.  Corrections per MI355X_MICROARCH.md (HBM / rocprofv3): the counters are in KiB, and on
    gfx950 FETCH_SIZE tallies the 128-B requests of a wide (16 B per lane) streaming read at 64 B: x2; WRITE_SIZE as is.
    Returns a dict (bytes per launch = median over the child's exact_kernel dispatches) or None if rocprofv3 is not usable."""
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None
    out = {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    kern_ms = []
    names = {}  # the dispatched kernels' names as the profiler saw them
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"ott_pmc_{counter}_", dir="/tmp")
        try:
            # the program itself behind `--` (no env / shell hop: the profiler's preloaded library has the GPU initialised already)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "4", "--warmup", "1", "--rows", str(args.rows), "--dim", str(args.dim), "--k", str(args.k),
                   "--seed", str(args.seed), "--no-cpu-baseline", "--no-extras", "--traffic", "off"]
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150)
            if r.returncode != 0:
                return None
            vals = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as f:
                    for row in csv.DictReader(f):
                        if "exact_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            vals.append(float(row["Counter_Value"]))
                            names[row["Kernel_Name"]] = names.get(row["Kernel_Name"], 0) + 1
                            if "End_Timestamp" in row and counter == "FETCH_SIZE":
                                kern_ms.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
            if not vals:
                return None
            out[counter] = float(np.median(vals))
            out[counter + "_dispatches"] = len(vals)
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"bytes": out["FETCH_SIZE"] * 1024 * 2 + out["WRITE_SIZE"] * 1024,
            "source": f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (median of {out['FETCH_SIZE_dispatches']} exact_kernel dispatches)",
            "correction": "KiB x 1024; FETCH_SIZE x 2 (gfx950 tallies a 128-B streaming request at 64 B); WRITE_SIZE as is",
            "kernel_ms_under_pmc": round(float(np.median(kern_ms)), 4) if kern_ms else None,
            "kernel_observed": max(names, key=names.get) if names else None}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0x07735)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational measurements beside the headline (N = 1: config 2; N > 1: config 4's shape and the strong-scaling split)")
    ap.add_argument("--c4-queries", type=int, default=1024, help="N > 1 extras: queries per batch of the config-4 shape")
    ap.add_argument("--inprocess", action="store_true",
                    help="N GPUs inside THIS one process: one store over all of them (ott_store_create_multi; VecStore(devices=[0..N-1])), "
                         "the reference's own single-process shape.  Default for N > 1 (what the driver launches): one process per GPU")
    ap.add_argument("--traffic", choices=("live", "profile", "off"), default="live",
                    help="roofline.traffic: 'live' = PMC child runs of this command under rocprofv3 (N = 1 only; falls back to "
                         "'profile'), 'profile' = the newest committed profiles/roundN/bench_n1_pmc_*.csv, 'off' = null")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.inprocess and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--inprocess is ONE process over N GPUs: do not start it under a multi-rank launcher")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.inprocess:
        return launch_ranks(args)  # parent: has made no HIP / torch call and makes none

    # stdout carries the ONE JSON line and nothing else: libraries loaded below print there too (RCCL's version banner goes to
    # the C stdout and is flushed at exit, i.e. BEHIND the line), so the real stdout is kept aside for the line and file
    # descriptor 1 is pointed at stderr for everybody else
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    live_traffic = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    traffic_attempt = None
    if args.traffic == "live" and world == 1 and not args.inprocess and os.path.exists("/dev/kfd") and not under_profiler:
        t_tr = time.perf_counter()
        live_traffic = measure_traffic_live(args)  # child processes; this process has not touched a GPU yet
        traffic_attempt = {"seconds": round(time.perf_counter() - t_tr, 1), "ok": live_traffic is not None}
    n_shards = args.gpus if args.inprocess else 1  # shards inside this process
    if args.gpus != world and not args.inprocess:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start the ranks with `python bench.py --gpus N` "
                         f"or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")

    import torch
    from otters_amd import Metric, Path, VecStore
    from otters_amd import _native as N
    from otters_amd.dist import Comm, ShardedVecStore

    # OTT_BENCH_SINGLE_DEVICE=1: every rank uses GPU 0 and the candidate blocks travel over the host transport (gloo) — a
    # functional check of the N-rank path on a 1-GPU box (RCCL refuses two ranks on one device); never a performance figure
    # OTT_BENCH_SINGLE_DEVICE=rccl: every rank uses GPU 0 AND the blocks travel through RCCL: each rank names itself a host of
    # its own (NCCL_HOSTID), so RCCL sees `world` one-GPU nodes and connects them through its socket transport over `lo` — the
    # real ncclCommInitRank(world) / ncclAllGather between RCCL kernels, minus xGMI.  Functional evidence only, like "1".
    single_mode = os.environ.get("OTT_BENCH_SINGLE_DEVICE", "")
    single_dev = single_mode == "1"
    single_rccl = single_mode == "rccl"
    if single_dev or single_rccl:
        local_rank = 0
    if single_rccl:
        os.environ["NCCL_HOSTID"] = f"ott-bench-rank-{rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    n_dev = C_int_device_count(N)
    if (world > n_dev and not (single_dev or single_rccl)) or local_rank >= n_dev:  # every rank sees the same shortfall and stops before any rendezvous
        raise SystemExit(f"[bench] --gpus {args.gpus} needs {world} GPUs (rank {rank} -> GPU {local_rank}) but this machine has {n_dev}")
    if args.inprocess and n_shards > n_dev and not single_mode:
        raise SystemExit(f"[bench] --gpus {args.gpus} --inprocess needs {n_shards} GPUs but this machine has {n_dev} "
                         f"(OTT_BENCH_SINGLE_DEVICE=1 puts every shard on GPU 0: a functional check, never a performance figure)")

    dist = None
    comm = None
    # OTT_BENCH_FORCE_DIST=1 runs the sharded code path (ott_query_sharded over a 1-rank RCCL comm) with one rank
    if world > 1 or os.environ.get("OTT_BENCH_FORCE_DIST") == "1":
        if world > 1:
            # control plane only (carries the 128-byte RCCL id, the barriers and the max-over-ranks of the timing)
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # data plane: RCCL behind the C ABI.  No fallback: a failure to create the communicator ends the run (non-zero)
        if single_dev:
            comm = Comm.from_torch(dist, local_rank, transport="host")
        elif world > 1:
            comm = Comm.from_torch(dist, local_rank, transport="rccl")
        else:
            comm = Comm.rccl(Comm.unique_id(), 0, 1, local_rank)
        probe = comm.all_gather_host(np.array([rank], dtype=np.int64)).ravel().tolist()  # fail here, not in the timed loop
        if probe != list(range(world)):
            raise SystemExit(f"[bench] the {comm.transport} communicator returned {probe}")

    if args.inprocess:
        # ONE store over N GPUs of this process: the same calls as on one GPU (reserve plans the even split over the shards)
        devices = [0] * n_shards if single_mode else list(range(n_shards))
        store = VecStore(args.dim, devices=devices)
        store.reserve(n_shards * args.rows)
        store.append_random(n_shards * args.rows, args.seed)
        layout = store.shards()
        # (shard boundaries fall on chunk boundaries: a shard holds args.rows rows give or take one 1024-row chunk)
        if (len(layout) != n_shards or any(abs(c - args.rows) > 1024 for _, _, c in layout) or sum(c for _, _, c in layout) != n_shards * args.rows
                or sorted({d for d, _, _ in layout}) != sorted(set(devices))):
            raise SystemExit(f"[bench] the in-process store's shards are {layout}, expected {n_shards} x ~{args.rows} rows on devices {devices}")
    else:
        store = VecStore(args.dim, device=local_rank)
        store.set_base_offset(rank * args.rows)
        store.reserve(args.rows)
        store.append_random(args.rows, args.seed)
    sharded = ShardedVecStore(store, comm, global_rows=world * args.rows) if comm is not None else None
    # Loading is over when the store's background plane builder is (option hi_prebuild: it converts the rows to the batch path's int8
    # plane right after the appends, ~30 ms of kernels at 10M x 768).  Left running, it overlaps the warm-up and — under a profiler,
    # whose start-up shifts everything — the first timed steps: the round-6 kernel trace showed five 5.1-5.6 ms dispatches among
    # twenty-one of 4.37.  The timed region measures queries on a loaded store, so the load is waited for (at most 10 s) out here.
    t_wait = time.perf_counter()
    while not args.inprocess and args.rows >= 262144 and not store.batch_ready() and time.perf_counter() - t_wait < 10.0:  # (smaller stores: no background build)
        time.sleep(0.005)

    rng = np.random.default_rng(args.seed + 1)
    queries = rng.uniform(-1, 1, (args.steps + args.warmup, args.dim)).astype(np.float32)

    # The headline is the reference's own arithmetic: the exact-order f32 kernel (src/vec_compute.rs:9-32), asked for by name.
    # AUTO would answer a single query on a store this size through the 16-bit cascade once its plane is resident — which it now
    # is soon after loading (option hi_prebuild): same bits, half the bytes (extras.single_query_via_cascade_ms) — but then
    # "GB/s scanned" would no longer be bytes that were read.
    def run(q):
        if sharded is not None:
            return sharded.query(q, Metric.Cosine).take(args.k).with_path(Path.Exact).collect_arrays()[0]
        return store.query(q, Metric.Cosine).take(args.k).with_path(Path.Exact).collect_arrays()[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(local_rank)
        N.check(N.lib().ott_store_sync(store._handle()))

    # ---- parity gate: nothing is timed before the GPU result has been checked on this run ----------
    parity = {}
    t_par = time.perf_counter()
    hits0 = run(queries[0])
    if hits0.size != args.k:
        raise SystemExit(f"[bench] PARITY FAILED: {hits0.size} hits, expected {args.k}")
    parity_rescore(hits0, queries[0], args.dim, args.seed)  # (a) every returned score, bit for bit, from regenerated rows
    parity["rescored_by_oracle"] = int(hits0.size)
    sample = None
    if world == 1 and not args.inprocess and not args.no_cpu_baseline and args.rows >= SAMPLE_ROWS:
        # (b) the same query restricted (zonemap-style chunk mask) to the rows the CPU baseline scores anyway: the whole
        # top-k — indices, order and score bits — against the oracle's answer over those rows
        import oracle as O
        sample = cpu_sample(args.dim, args.seed)
        n_s = sample[0].shape[0]
        n_chunks = (args.rows + 1023) // 1024
        cmask = np.zeros(n_chunks, dtype=bool)
        cmask[: n_s // 1024] = True
        rq = store.query(queries[0], Metric.Cosine).take(args.k).with_path(Path.Exact).resolve()
        got, _, _ = store._run(rq, chunk_mask=cmask)
        ref = O.vec_query(sample[0], queries[0], O.METRIC_COSINE, O.TAKE_MAX, args.k, inv=sample[1], ties=O.TIES_CANONICAL)
        if not (np.array_equal(got["index"], ref["index"]) and np.array_equal(bits(got["score"]), bits(ref["score"]))):
            raise SystemExit(f"[bench] PARITY FAILED: top-{args.k} over the first {n_s} rows differs from the oracle\n got {got}\n ref {ref}")
        parity["topk_vs_oracle_rows"] = int(n_s)
    parity["seconds"] = round(time.perf_counter() - t_par, 2)

    for i in range(args.warmup):
        run(queries[i])
    # The interpreter's cyclic garbage collector stays out of the timed region (as `timeit` keeps it out): with torch imported a full
    # collection walks several hundred thousand objects and takes ~40 ms — one step of a 10-step run measured 39-45 ms instead of 4.4,
    # always the same step (the allocation count that triggers the collection is deterministic), until this was found in round 6.
    gc.collect()
    gc.disable()
    barrier()
    kernel_ns, exchange_ns, merge_ns = [], [], []
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = run(queries[args.warmup + i])
        kernel_ns.append(store.last_stats["score_ns"])  # hipEvent time of the scoring kernel on the store's stream (in-process store: the slowest shard's)
        exchange_ns.append(store.last_stats["exchange_ns"])  # N > 1: end of own scoring -> start of the cross-GPU merge
        merge_ns.append(store.last_stats["merge_ns"])
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if res.size != args.k:
        raise SystemExit(f"[bench] {res.size} hits in the timed loop, expected {args.k}")
    if store.last_stats["path_used"] != int(Path.Exact):  # the headline is the exact-order f32 kernel, never the bf16 cascade
        raise SystemExit(f"[bench] the timed loop ran on path {store.last_stats['path_used']}, not on the exact-order kernel")

    rank_kernel_ms = [float(np.mean(kernel_ns)) / 1e6]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allk = torch.zeros(world, dtype=torch.float64)
        dist.all_gather_into_tensor(allk, torch.tensor(rank_kernel_ms, dtype=torch.float64))
        rank_kernel_ms = [float(x) for x in allk]
    # N > 1: what the transport itself says about the job (the day the driver has 8 GPUs the line proves which exchange ran)
    comm_info = comm.info() if comm is not None else None
    if comm is not None and comm.transport == "rccl" and comm_info["nranks"] != args.gpus:
        raise SystemExit(f"[bench] the RCCL communicator spans {comm_info['nranks']} ranks, --gpus is {args.gpus}")

    # N > 1 (either launcher): what the 8-GPU node must answer BESIDE the weak-scaling headline — config 4's shape and the metric's
    # own corpus split N ways.  Collective: every rank runs it, rank 0 reports it.  Never inside the timed region above.
    multi_extras = None
    if (world > 1 or args.inprocess or comm is not None) and not args.no_extras:
        try:
            multi_extras = multi_gpu_extras(args, store, sharded, comm, world, n_shards, rng, Metric, Path, barrier, dist, torch)
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001 -- a failure here must not hang the peers of a collective: every rank fails the same way or none
            multi_extras = {"error": repr(e)}

    if rank == 0:
        bytes_per_pass = args.rows * (args.dim * 4 + 4)  # algorithmic: 4*dim per row + 4 B inverse norm (cosine)
        ms_per_step = dt / args.steps * 1e3
        qps = args.steps / dt
        gbs = world * n_shards * bytes_per_pass * qps / 1e9
        kern_ms = float(np.mean(kernel_ns)) / 1e6
        achieved = bytes_per_pass / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch of the headline kernel (separate --pmc passes; FETCH_SIZE is in KiB and counts half the bytes of
        # a wide streaming read on gfx950: x2; WRITE_SIZE as is): measured by this run's own PMC child runs (--traffic live),
        # else read from the newest committed profile of this same command — `traffic_source` says which
        traffic, traffic_src, traffic_note = None, None, None
        if live_traffic is not None:
            traffic, traffic_src = live_traffic["bytes"], live_traffic["source"]
            traffic_note = {"correction": live_traffic["correction"], "kernel_ms_under_pmc": live_traffic["kernel_ms_under_pmc"],
                            "live_attempt": traffic_attempt}
        elif args.traffic != "off" and (args.rows, args.dim) == (10_000_000, 768) and world == 1 and not args.inprocess:
            f_kib, f_src = profile_counter("FETCH_SIZE", "exact_kernel")
            w_kib, _ = profile_counter("WRITE_SIZE", "exact_kernel")
            if f_kib is not None:
                traffic, traffic_src = f_kib * 1024 * 2 + (w_kib or 0.0) * 1024, f"committed profile, NOT this run: {f_src}"
                traffic_note = {"correction": "KiB x 1024; FETCH_SIZE x 2 (gfx950); WRITE_SIZE as is", "live_attempt": traffic_attempt}
        # the kernel the timed loop launched: exact_kernel<L2, NQ, E, PERQ, DUMP, SMALL, BLK> (ott_exact.hip) for this metric / k / size
        # (a single query takes the register lists up to k = 512, the sort path beyond; stores of up to 1024 tiles: rows8)
        n_tiles = (args.rows + 63) // 64
        e_lane = 1 if args.k <= 64 else 2 if args.k <= 128 else 4 if args.k <= 256 else 8
        blk = "true" if (e_lane > 1 or args.k > 16) else "false"
        if args.k > 512:
            kernel_name = "ott::exact_kernel<false, 1, 1, false, true, false, false> (score dump, two phases) + radix sort"
        elif n_tiles <= 1024 and e_lane <= 2 and args.dim <= 2048:
            kernel_name = f"ott::exact_rows8_kernel<false, {e_lane}, 1, false>"
        else:
            kernel_name = f"ott::exact_kernel<false, 1, {e_lane}, false, false, false, {blk}>"
        kernel_src = "derived from k / rows (ott_exact.hip's dispatch rule)"
        if live_traffic is not None and live_traffic.get("kernel_observed"):
            kernel_name, kernel_src = live_traffic["kernel_observed"], "observed: the dispatch name rocprofv3 recorded in this run's PMC child"
        sharding = "none"
        if args.inprocess:
            sharding = (f"in-process: ONE store over {n_shards} GPUs of one process (ott_store_create_multi), shards of {args.rows} rows, "
                        f"{store.transport()} exchange of per-GPU top-{args.k} + device merge on the first GPU"
                        + (" (every shard on GPU 0: functional check only)" if single_mode else ""))
        if comm is not None:
            sharding = (f"{world} row shards, ott_query_sharded: {comm.transport.upper()} all-gather of per-GPU top-{args.k} + device merge"
                        + (" (host transport over gloo: functional check only)" if comm.transport != "rccl" else "")
                        + (" (all ranks on GPU 0, RCCL's socket transport between them: functional check only)" if single_rccl else ""))
        line = {
            "metric": "GB/s scanned + queries/sec, exact cosine top-10 over 10M x 768 f32 rows per GPU",
            "value": round(gbs, 2), "unit": "GB/s", "queries_per_sec": round(qps, 2),
            "n_gpus": world * n_shards, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "parity_checked": True, "parity": parity,
            "config": {"workload": f"{args.rows}x{args.dim} f32 VecStore per GPU, single query, Metric::Cosine, take({args.k})",
                       "rows_per_gpu": args.rows, "dim": args.dim, "k": args.k, "nq": 1,
                       "sharding": sharding, "processes": world,
                       "transport": comm.transport if comm is not None else (store.transport() if args.inprocess else None),
                       "path": "exact-order VALU scorer + fused wavefront top-k (Path.Exact, asked for by name)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
                         "kernel": kernel_name, "kernel_source": kernel_src, "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": bytes_per_pass},
        }
        if comm is not None or args.inprocess:
            line["exchange"] = {
                "rccl": ({"nranks": comm_info["nranks"], "version": comm_info["version"]} if comm is not None and comm.transport == "rccl" else None),
                "allgather_us": round(float(np.median(exchange_ns)) / 1e3, 1),  # rank 0 / the merging GPU: own scoring done -> merge starts
                "merge_us": round(float(np.median(merge_ns)) / 1e3, 1),         # block-list merge + cross-GPU merge kernels
                "kernel_ms_per_rank": {"min": round(min(rank_kernel_ms), 4), "max": round(max(rank_kernel_ms), 4)},
            }
        if multi_extras is not None:
            line["extras"] = multi_extras
        if world == 1 and comm is None and not args.no_extras and not args.inprocess:
            try:
                line["extras"] = config2_extras(store, rng, args, queries, Metric, Path)
            except SystemExit:
                raise
            except Exception as e:  # noqa: BLE001 -- never let the informational part break the contract line
                line["extras"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1 and not args.inprocess:
            if sample is None:
                sample = cpu_sample(args.dim, args.seed)
            line["cpu_baseline"] = cpu_baseline(sample[0], sample[1], queries[:1], args.k)
            if "extras" in line and isinstance(line["extras"], dict):
                line["extras"]["cpu_baseline_all_cores_config3"] = cpu_baseline_all_cores(store, sample, queries[:1], args.k)
        line_out.write(json.dumps(line) + "\n")
        line_out.flush()

    if dist is not None:
        dist.barrier()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def C_int_device_count(N) -> int:
    import ctypes as C
    n = C.c_int(0)
    N.check(N.lib().ott_device_count(C.byref(n)))
    return n.value


def multi_gpu_extras(args, store, sharded, comm, world, n_shards, rng, Metric, Path, barrier, dist, torch) -> dict:
    """N > 1: the two measurements the weak-scaling headline does not carry (VERDICT r5 missing #1), both over the rows already
    resident (a chunk mask selects each GPU's share; nothing is re-loaded):

    config4     BASELINE config 4's shape: 40M x 768 over 8 GPUs = 5M rows per GPU (here: min(5M, half of --rows) per GPU; all of them when --rows <= 5M), a batch
                of 1024 queries, cosine, take(100) — merged (the reference's one list over all (query, row) pairs, src/vec.rs:217-219)
                and per query — through the default path (int8 level first) with the per-GPU candidate blocks all-gathered and merged
                (src/meta.rs:678-709 is what the exchange stands for); parity: 8 sampled queries against the exact-order path.
    strong_10M  the metric's own corpus, --rows rows IN TOTAL split N ways (rows / N per GPU), single query, cosine top-k on the
                exact-order kernel: the fixed-corpus ("strong") reading of "1/2/4/8 GPU" beside the headline's weak one."""
    CS = 1024
    n_gpus = world * n_shards
    inproc = sharded is None
    layout = store.shards() if (inproc and n_shards > 1) else None  # [(device, first row, rows)] of the in-process store

    def mask_first(rows_per_gpu):
        """chunk mask (this rank's store, local chunk ids): the first `rows_per_gpu` rows of every GPU's shard"""
        if layout is None:
            n_chunks = (args.rows + CS - 1) // CS
            m = np.zeros(n_chunks, dtype=bool)
            m[: rows_per_gpu // CS] = True
            return m
        total = sum(c for _, _, c in layout)
        m = np.zeros((total + CS - 1) // CS, dtype=bool)
        for _, first, _cnt in layout:
            m[first // CS: first // CS + rows_per_gpu // CS] = True
        return m

    def run_rq(rq, cmask):
        if sharded is not None:
            return sharded._run(rq, chunk_mask=cmask)[0]  # (ott_query_sharded; leaves its stats in store.last_stats)
        hits, _, stats = store._run(rq, chunk_mask=cmask)
        store.last_stats = stats
        return hits

    def timed(fn, warm, steps):
        for _ in range(warm):
            fn()
        gc.collect()
        gc.disable()  # (as in the headline's timed region: a full collection costs ~40 ms with torch imported)
        barrier()
        sc, ex, mg = [], [], []
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
            st = store.last_stats
            sc.append(st["score_ns"]); ex.append(st["exchange_ns"]); mg.append(st["merge_ns"])
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / steps, float(np.median(sc)) / 1e6, float(np.median(ex)) / 1e3, float(np.median(mg)) / 1e3

    out = {}
    # ---- config 4 -----------------------------------------------------------------------------------------------------------------
    rows_c4 = (min(5_000_000, args.rows if args.rows <= 5_000_000 else args.rows // 2) // CS) * CS  # (--rows 5000000: the whole shard IS config 4's share)
    nq, k = int(args.c4_queries), 100
    if rows_c4 >= CS and nq >= 8:
        cm = mask_first(rows_c4)
        qs = rng.uniform(-1, 1, (nq, args.dim)).astype(np.float32)
        source = sharded if sharded is not None else store
        c4 = {"workload": f"{n_gpus} x {rows_c4} rows x {args.dim} f32 ({n_gpus * rows_c4} rows in all; BASELINE config 4 is 8 x 5M), {nq}-query batch, "
                          f"Metric::Cosine, take({k}); each GPU's share = the first {rows_c4} rows of its resident shard (chunk mask)",
              "rows_per_gpu": rows_c4, "n_gpus": n_gpus, "nq": nq, "k": k}
        for mode, perq in (("merged", False), ("per_query", True)):
            plan = source.query(qs, Metric.Cosine).take(k)
            if perq:
                plan = plan.per_query()
            rq = plan.resolve()
            last = [None]

            def batch(rq=rq, last=last):
                last[0] = run_rq(rq, cm)
            sec, score_ms, ex_us, mg_us = timed(batch, 2, 5)
            hits = last[0]
            st = dict(store.last_stats)
            # candidate block every GPU contributes to the exchange: merged k hits, per query nq x k hits, 16 B each (ott_hit)
            block = (nq * k if perq else k) * 16
            c4[mode] = {"ms_per_batch": round(sec * 1e3, 3), "queries_per_sec": round(nq / sec, 1), "score_phase_ms": round(score_ms, 3),
                        "allgather_us": round(ex_us, 1), "merge_us": round(mg_us, 1), "exchange_bytes_per_gpu": block,
                        "exchange_bytes_all_gpus": block * n_gpus, "hits": int(hits.size), "path_used": st["path_used"],
                        "queries_refined_int8_level": st["i8_refined"], "queries_rerun_exact": st["retries"]}
            if perq:
                # parity: 8 sampled queries of the batch, exact-order path, same rows: indices, order and score bits
                pick = np.linspace(0, nq - 1, 8).astype(np.int64)
                rq_e = source.query(qs[pick], Metric.Cosine).take(k).per_query().with_path(Path.Exact).resolve()
                ref = run_rq(rq_e, cm)
                got = hits.reshape(nq, k)[pick].reshape(-1)
                if not (np.array_equal(got["index"], ref["index"]) and np.array_equal(bits(got["score"]), bits(ref["score"]))):
                    raise SystemExit("[bench] PARITY FAILED: config-4 batch (default path) differs from the exact-order path on the sampled queries")
                c4["parity_checked_queries"] = int(pick.size)
        if comm is not None:
            info = comm.info()
            c4["rccl"] = {"nranks": info["nranks"], "version": info["version"]} if comm.transport == "rccl" else None
            c4["transport"] = comm.transport
        elif inproc:
            c4["transport"] = store.transport() if n_shards > 1 else None
        out["config4"] = c4
    # ---- the metric's corpus split N ways ---------------------------------------------------------------------------------------------
    rows_s = ((args.rows // n_gpus) // CS) * CS
    if rows_s >= CS:
        cm = mask_first(rows_s)
        source = sharded if sharded is not None else store
        q1 = rng.uniform(-1, 1, (8, args.dim)).astype(np.float32)
        rqs = [source.query(q1[i], Metric.Cosine).take(args.k).with_path(Path.Exact).resolve() for i in range(8)]
        it = [0]

        def one():
            run_rq(rqs[it[0] % 8], cm)
            it[0] += 1
        sec, score_ms, ex_us, mg_us = timed(one, 3, max(args.steps, 5))
        out["strong_10M"] = {
            "workload": f"{n_gpus * rows_s} rows x {args.dim} f32 in all, {rows_s} per GPU (the first rows of each resident shard), single query, "
                        f"Metric::Cosine, take({args.k}), exact-order kernel",
            "scaling": "strong", "rows_total": n_gpus * rows_s, "rows_per_gpu": rows_s, "n_gpus": n_gpus,
            "ms_per_step": round(sec * 1e3, 4), "queries_per_sec": round(1.0 / sec, 2),
            "GBs_scanned": round(n_gpus * rows_s * (args.dim * 4 + 4) / sec / 1e9, 2),
            "kernel_ms": round(score_ms, 4), "allgather_us": round(ex_us, 1), "merge_us": round(mg_us, 1)}
    out["note"] = "value / scaling of the line stay the weak-scaling headline (rows PER GPU fixed); config4 and strong_10M are labelled readings beside it"
    return out


def config2_extras(store, rng, args, queries, Metric, Path) -> dict:
    """Outside the timed region, informational: BASELINE config 2 (256 queries, Metric::Cosine, take(100), merged = the
    reference's semantics) on the same resident corpus through the batch path (certified cascade: 16-bit hi-plane candidate
    pass on the matrix cores — IEEE half since round 3 —, split-bf16 pass for what it cannot certify, exact f32 re-score of
    every candidate), parity checked against the exact-order kernel, with the roofline of its dominant kernel: the hi pass
    streams the 16-bit plane once (255 flop/B against a 16-bit matrix balance of 312) — both fractions are reported, and the
    f32-pipe variant's beside them."""
    nq, k = 256, 100
    Q = rng.uniform(-1, 1, (nq, args.dim)).astype(np.float32)
    plane_ready = store.batch_ready()  # the background build after the appends (option hi_prebuild) has finished
    t_first = time.perf_counter()
    got, _ = store.query(Q, Metric.Cosine).take(k).collect_arrays()  # the first batch on this store (whatever it still has to build rides here)
    first_batch_ms = (time.perf_counter() - t_first) * 1e3
    # parity: the merged top-100 over all 256 x rows pairs against the exact-order kernel (64 four-query passes), and four
    # per-query lists likewise — indices, order, query ids, score bits
    ref, _ = store.query(Q, Metric.Cosine).take(k).with_path(Path.Exact).collect_arrays()
    same = (np.array_equal(got["index"], ref["index"]) and np.array_equal(got["query"], ref["query"])
            and np.array_equal(bits(got["score"]), bits(ref["score"])))
    pick = [0, 85, 170, 255]
    pq, cnt = store.query(Q, Metric.Cosine).take(k).per_query().collect_arrays()
    pr, _ = store.query(Q[pick], Metric.Cosine).take(k).per_query().with_path(Path.Exact).collect_arrays()
    for j, qi in enumerate(pick):
        a, b = pq[qi * k:(qi + 1) * k], pr[j * k:(j + 1) * k]
        same = same and np.array_equal(a["index"], b["index"]) and np.array_equal(bits(a["score"]), bits(b["score"]))
    if not same or cnt != [k] * nq:
        raise SystemExit("[bench] PARITY FAILED: config-2 batch (256 queries, top-100) differs from the exact-order kernel")
    reps = 7
    score_ms, wall_ms = [], []
    for _ in range(reps):
        t1 = time.perf_counter()
        store.query(Q, Metric.Cosine).take(k).collect_arrays()
        wall_ms.append((time.perf_counter() - t1) * 1e3)
        score_ms.append(store.last_stats["score_ns"] / 1e6)
    bdt = float(np.median(wall_ms)) / 1e3  # median of 7 batches (a single host hiccup used to move the mean by a millisecond)
    st = store.last_stats
    sms = float(np.median(score_ms))
    flops = 2.0 * args.rows * args.dim * nq
    # the plane the first candidate pass streams: since round 5 the int8 plane (row pitch = dim rounded to 128 bytes) + one f32
    # scale and one inverse norm per row
    plane_bytes = args.rows * ((args.dim + 127) // 128 * 128) + args.rows * 8
    busy, busy_src = profile_mfma_busy()
    ex = {"first_batch_ms": round(first_batch_ms, 3), "hi_plane_ready_before_first_batch": bool(plane_ready),
          "config2_256q_top100_ms_per_batch": round(bdt * 1e3, 3),
          "config2_queries_per_sec": round(nq / bdt, 1),
          "config2_score_phase_ms": round(sms, 3),
          "config2_queries_refined_split_pass": int(st["refined"]),
          "config2_queries_rerun_exact": int(st["retries"]),
          "config2_parity_checked": True,
          "config2_queries_refined_int8_level": int(st["i8_refined"]),
          "config2_roofline": {"bound": "mfma", "kernel": "int8 pass (per-row-scaled int8 plane, v_mfma_i32_32x32x32_i8: exact i32 accumulation), 256-query tile; "
                                                          "512 candidates per query re-scored in the reference's f32 order",
                               "plane_bytes": plane_bytes, "score_phase_ms": round(sms, 3),
                               "achieved_GBs": round(plane_bytes / (sms * 1e-3) / 1e9, 1),
                               "frac_hbm": round(plane_bytes / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "int8_tops": round(flops / (sms * 1e-3) / 1e12, 1),
                               "frac_int8_peak": round(flops / (sms * 1e-3) / 1e12 / I8_PEAK_TOPS, 4),
                               "mfma_busy": busy, "mfma_busy_source": busy_src,
                               # neither peak above is the roof that binds: the tile's own fill pattern (rows from HBM + the query block from
                               # L2 through the L2 -> LDS fill), measured alone, takes 1.58 ms per batch of this shape — the two streams add up
                               # (profiles/round6/fill_path.md, benchmarks/cpp/fill_path.hip: a committed measurement, not taken in this run)
                               "fill_floor": {"ms_per_batch": 1.58, "source": "profiles/round6/fill_path.md (committed; not measured in this run)",
                                              "score_phase_over_floor": round(sms / 1.58, 3),
                                              "note": "K loops 1.64-1.68 ms = 0.95 of the floor (in-kernel stamps, round 5); the rest of the score phase is epilogue and rounds"}
                                              if (args.rows, args.dim) == (10_000_000, 768) else None}}
    # the f32 matrix pipe (v_mfma_f32_32x32x2_f32: north_star's ">= 40 % MFMA peak" read literally), same batch, same run
    store.set_option("mfma_f32", 1)
    try:
        g32, _ = store.query(Q, Metric.Cosine).take(k).with_path(Path.Mfma).collect_arrays()
        if not (np.array_equal(g32["index"], ref["index"]) and np.array_equal(bits(g32["score"]), bits(ref["score"]))):
            raise SystemExit("[bench] PARITY FAILED: config-2 batch on the f32 matrix pipe differs from the exact-order kernel")
        f_ms = []
        for _ in range(3):
            store.query(Q, Metric.Cosine).take(k).with_path(Path.Mfma).collect_arrays()
            f_ms.append(store.last_stats["score_ns"] / 1e6)
        fm = float(np.median(f_ms))
        ex["config2_roofline"]["f32_pipe"] = {"bound": "mfma", "score_phase_ms": round(fm, 3), "tflops": round(flops / (fm * 1e-3) / 1e12, 1),
                                              "peak": F32_MFMA_PEAK_TFLOPS, "frac_f32_mfma_peak": round(flops / (fm * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
    finally:
        store.set_option("mfma_f32", 0)
    # opt-in, also informational: ONE query through the same cascade (Path.Mfma) instead of the exact-order kernel
    # the timed region above measures (AUTO keeps single queries on that kernel: no second copy of the corpus)
    store.query(queries[0], Metric.Cosine).take(args.k).with_path(Path.Mfma).collect()
    t1 = time.perf_counter()
    for i in range(10):
        store.query(queries[i % len(queries)], Metric.Cosine).take(args.k).with_path(Path.Mfma).collect()
    ex["single_query_via_cascade_ms"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
    return ex


def profile_mfma_busy():
    """Busy fraction of the matrix pipe in the batch path's largest candidate-pass dispatch, from a committed rocprofv3 PMC
    pass (profiles/round5/i8_pmc/c2_i8_pmc_MFMA_BUSY.csv, benchmarks/profile_i8.sh; before round 5 profiles/roundN/c2_hi_pmc_MFMA_BUSY.csv): SQ_VALU_MFMA_BUSY_CYCLES /
    (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs).  (None, None) when no such profile is present."""
    pdir = os.path.join(ROOT, "profiles")
    rounds = sorted((d for d in os.listdir(pdir) if d.startswith("round")), reverse=True) if os.path.isdir(pdir) else []
    for rd in rounds:
        path = os.path.join(pdir, rd, "i8_pmc", "c2_i8_pmc_MFMA_BUSY.csv")
        if not os.path.exists(path):
            path = os.path.join(pdir, rd, "c2_hi_pmc_MFMA_BUSY.csv")
        if not os.path.exists(path):
            continue
        per, dur = {}, {}
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if "mfma_score" in row["Kernel_Name"] or "hi256" in row["Kernel_Name"]:
                    d = per.setdefault(row["Dispatch_Id"], {})
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        if dur:
            x = per[max(dur, key=dur.get)]
            if x.get("GRBM_GUI_ACTIVE") and x.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                return round(x["SQ_VALU_MFMA_BUSY_CYCLES"] / (x["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 4), os.path.relpath(path, ROOT)
    return None, None


if __name__ == "__main__":
    sys.exit(main())
