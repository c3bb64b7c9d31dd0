"""Adversarial corpora for the int8 level's error bound (test infrastructure; numpy only, no oracle, no GPU).

The int8 candidate passes (run_mfma level 2 and the single-query sweep, otters_amd/csrc/ott_mfma.hip) accumulate in exact
integers, so their approximate score sits within a few 2^-24 of the REAL dot product whenever rows and queries are
int8-representable (row = k * s, k integer in [-127, 127] with a +-127 in every row): the measured quantisation losses are ~1e-7.
The exact-order f32 re-score (the reference's arithmetic, /root/reference/src/vec_compute.rs:9-22: eight lane chains of dim/8
rounded products and rounded adds, reduce_add, remainder) is NOT: nearly constant rows add (almost) the same product again and
again, every add rounds the same way, and the lane sums drift systematically by tens of 2^-24 relative.  A bound on
|approximate - exact-order| that prices only the approximate side (16 * 2^-24, the round-5 constant) is therefore not a bound.

`build_case` constructs a corpus that turns that into a wrong answer under such a bound:

  * F rows (>= 600 identical copies at the lowest indices): fill the candidate list; their approximate score is U, the list's cut;
  * P rows (k copies): approximate and exact score a little above U + eps_old — the list's exact top-k, "certified";
  * A rows (k copies at the highest indices): approximate score just BELOW U (never listed), exact-order score ABOVE the P rows'
    (the drift): the reference returns the A rows;
  * a neighbourhood of near ties a few 2^-24 apart below the A rows, and random int8-representable filler.

Everything here is emulation in numpy float32 (IEEE, no fused multiply-add), used to CHOOSE the rows and to state the
preconditions; the tests' verdict is the oracle's (tests/test_gpu_i8_bound.py) — this module never decides what is correct."""
import numpy as np

f32 = np.float32
U24 = 2.0 ** -24


def seq_sumsq(x: np.ndarray) -> np.ndarray:
    """sequential f32 sum of squares per row (src/vec.rs:387-397)"""
    s = np.zeros(x.shape[0], f32)
    for j in range(x.shape[1]):
        s = s + x[:, j] * x[:, j]
    return s


def inv_norms(x: np.ndarray) -> np.ndarray:
    return (f32(1) / np.sqrt(seq_sumsq(x))).astype(f32)


def exact_dot(rows: np.ndarray, q: np.ndarray) -> np.ndarray:
    """the reference's dot in its order of operations, vectorised over rows (src/vec_compute.rs:9-22; AVX reduce order)"""
    n, dim = rows.shape
    full = dim // 8
    acc = np.zeros((n, 8), f32)
    for s in range(full):
        acc = acc + rows[:, 8 * s:8 * s + 8] * q[8 * s:8 * s + 8]
    dot = ((acc[:, 0] + acc[:, 4]) + (acc[:, 2] + acc[:, 6])) + ((acc[:, 1] + acc[:, 5]) + (acc[:, 3] + acc[:, 7]))
    tail = np.zeros(n, f32)
    for j in range(8 * full, dim):
        tail = tail + rows[:, j] * q[j]
    return dot + tail


def exact_scores(rows, q, metric):
    """(exact-order f32 score, the same score with the dot taken in f64 — what an error-free dot would give through the SAME
    f32 normalisers)"""
    d = exact_dot(rows, q)
    rd = rows.astype(np.float64) @ q.astype(np.float64)
    if metric == "dot":
        return d, rd
    qi = inv_norms(q[None, :])[0]
    vi = inv_norms(rows)
    return (d * qi) * vi, rd * float(qi) * vi.astype(np.float64)


def i8_plane(rows):
    """the plane as i8_rows_kernel builds it (ott_store.hip): s_v = max|v| / 127, element = rint(v * (1 / s_v)); and the measured
    relative loss per row"""
    mx = np.abs(rows).max(axis=1).astype(f32)
    s = (mx / f32(127)).astype(f32)
    inv = (f32(1) / s).astype(f32)
    t = np.clip(np.rint(rows * inv[:, None]), -127, 127).astype(np.int32)
    err = rows.astype(np.float64) - s.astype(np.float64)[:, None] * t
    rel = np.sqrt((err * err).sum(1) / (rows.astype(np.float64) ** 2).sum(1))
    return t, s, rel


def i8_query(q, metric):
    """the query operand as run_i8_single / the batch prepare quantise it; (ints, s_Q, measured loss x 1.0001)"""
    pf = inv_norms(q[None, :])[0] if metric == "cosine" else f32(1)
    xe = (q * pf).astype(f32)
    e_max = f32(np.abs(q).max()) * pf
    s_q = f32(e_max / f32(127))
    inv = f32(f32(1) / s_q)
    t = np.clip(np.rint(xe * inv), -127, 127).astype(np.int32)
    df = xe.astype(np.float64) - float(s_q) * t
    qrel = float(np.sqrt((df * df).sum() / (xe.astype(np.float64) ** 2).sum()) * 1.0001)
    return t, s_q, qrel


def i8_approx(rows, q, metric):
    """the int8 passes' approximate score: (float)acc * ((vinv * s_v) * s_Q) (ott_exact.hip, I8 branch; the tile's row factor)"""
    t, s_v, rel = i8_plane(rows)
    tq, s_q, qrel = i8_query(q, metric)
    acc = (t.astype(np.int64) @ tq.astype(np.int64)).astype(f32)
    if metric == "cosine":
        rf = ((inv_norms(rows) * s_v).astype(f32) * s_q).astype(f32)
    else:
        rf = (s_v * s_q).astype(f32)
    return (acc * rf).astype(f32), rel, qrel


def eps_old(dim, metric, i8_rel, qrel, q, rows):
    """round 5's certified bound of the int8 level: (16 * 2^-24 + measured losses) [x ||q|| max||v|| for dot]"""
    r = 1.001 * (1.0 + 0.015625) * i8_rel + 1.001 * qrel
    e = 16.0 * U24 + r
    if metric == "dot":
        e *= float(np.linalg.norm(q.astype(np.float64))) * float(np.linalg.norm(rows.astype(np.float64), axis=1).max())
    return e


def _lanes_to_row(lane, dim):
    """a row whose eight lane chains are all `lane` (position 8 j + l holds lane[j]); remainder elements = 127"""
    k = np.repeat(np.asarray(lane, np.int64), 8)
    return np.concatenate([k, np.full(dim - k.shape[0], 127, np.int64)])


def build_case(dim: int, metric: str, seed: int, k: int = 10, n: int = 50_000, signed: bool = False, copies: int = 640):
    """-> dict(rows, query, expect_rows (the A rows' indices), info).  Deterministic in its arguments."""
    rng = np.random.default_rng(seed)
    L = dim // 8
    best = None
    for trial in range(16):
        s = f32(0.37 / 127) if trial == 0 else f32(rng.uniform(0.0025, 0.0035))
        kq = np.full(dim, 127, np.int64) if metric == "cosine" else np.where(np.arange(dim) % 3 == 0, 126, 127).astype(np.int64)
        q = (kq.astype(f32) * s).astype(f32)
        pats = []
        for m in range(0, L + 1, max(1, L // 48)):
            for c in range(40, 128):
                lane = [127] * m + [c] * (L - m)
                if m == 0:
                    lane[-1] = 127
                pats.append(lane)
        K = np.stack([_lanes_to_row(p, dim) for p in pats])
        R = (K.astype(f32) * s).astype(f32)
        ex, ideal = exact_scores(R, q, metric)
        scale = 1.0 if metric == "cosine" else float(np.linalg.norm(q.astype(np.float64))) * np.linalg.norm(R.astype(np.float64), axis=1)
        dl = (ex - ideal) / scale
        i = int(np.argmax(dl))
        if best is None or dl[i] > best[0]:
            best = (float(dl[i]), s, kq, K[i].copy())
    drift, s, kq, kA = best
    q = (kq.astype(f32) * s).astype(f32)

    RA = (kA[None, :].astype(f32) * s).astype(f32)
    exA, _ = exact_scores(RA, q, metric)
    apA, relA, qrel = i8_approx(RA, q, metric)
    exA, apA = float(exA[0]), float(apA[0])

    # Quiet rows (no drift: independent random elements) whose score is near A's.  Stage 1: rows of iid uniform integers on a range
    # chosen so that the expected score matches A's (cosine: mean / rms; dot: the mean, with a narrow range so that no norm exceeds
    # A's — the dot bound scales with the store's largest norm); stage 2: the nearest such row with one to three (+1, -1) pairs of
    # steps, each of which moves the score by a few 2^-24 — the near-tie neighbourhood the rows are picked from.
    meanA, rmsA = float(kA.mean()), float(np.sqrt((kA.astype(np.float64) ** 2).mean()))
    if metric == "cosine":
        los = np.arange(-60, 121)
        mean = (los + 127) / 2.0
        var = ((127 - los + 1.0) ** 2 - 1.0) / 12.0
        lo = int(los[np.argmin(np.abs(mean / np.sqrt(mean * mean + var) - meanA / rmsA))])
        hi = 127
    else:
        w = int(min(20, 127 - meanA, meanA - 1))
        lo, hi = int(round(meanA)) - w, int(round(meanA)) + w
    n1 = max(1500, min(6000, 6_000_000 // dim))
    K1 = rng.integers(lo, hi + 1, (n1, dim))
    K1[np.arange(n1), rng.integers(0, dim, n1)] = 127
    ap1, _, _ = i8_approx((K1.astype(f32) * s).astype(f32), q, metric)
    base = K1[int(np.argmin(np.abs(ap1.astype(np.float64) - apA)))]
    n_pool = max(3000, min(12000, 12_000_000 // dim))
    KP = np.repeat(base[None, :], n_pool, 0)
    if metric == "cosine":
        for i in range(n_pool):
            for _ in range(int(rng.integers(1, 4))):
                a, b = rng.choice(dim, 2, replace=False)
                if KP[i, a] < 126 and KP[i, b] > -127 and KP[i, b] != 127:
                    KP[i, a] += 1
                    KP[i, b] -= 1
    else:
        # dot: the score is the integer sum(k_q k_v) times a constant; +1 where k_q = 127 and -1 where k_q = 126 moves it by ONE, about
        # 2^-23 of the bound's scale.  Net steps spread over what the construction needs on either side of A's score
        hi_pos, lo_pos = np.where((kq == 127) & (base < 126))[0], np.where((kq == 126) & (base < 126))[0]
        b_ap = float(ap1[int(np.argmin(np.abs(ap1.astype(np.float64) - apA)))])
        step = b_ap / float((base * kq).sum())
        t0 = int(round((apA - b_ap) / step))                 # brings the base row's score onto A's
        span = int(1.3 * (exA - apA) / step) + 40            # ... and the pool spreads around it
        cap = min(len(hi_pos), len(lo_pos)) - 1
        for i in range(n_pool):
            t = t0 + int(rng.integers(-span // 3, span + 1))
            mag = (abs(t) + cap - 1) // cap if t else 1      # steps per position (1 unless the base row sits far from A)
            cnt, rem = abs(t) // mag, abs(t) % mag
            a = rng.choice(hi_pos, cnt + 1, replace=False)
            b = rng.choice(lo_pos, cnt + 1, replace=False)
            sg = 1 if t > 0 else -1
            KP[i, a[:cnt]] += sg * mag
            KP[i, b[:cnt]] -= sg * mag
            KP[i, a[cnt]] += sg * rem
            KP[i, b[cnt]] -= sg * rem
        assert KP.min() >= 1 and np.abs(KP).max() <= 127
    RP = (KP.astype(f32) * s).astype(f32)
    exP, _ = exact_scores(RP, q, metric)
    apP, relP, _ = i8_approx(RP, q, metric)
    i8_rel = float(max(relP.max(), relA.max()))
    e_old = eps_old(dim, metric, i8_rel, qrel, q, np.concatenate([RP, RA]))
    unit = e_old / 20.0  # ~2^-24 of the bound's scale
    # F: approx a little above A's approx; P: exact above U + eps_old, below A's exact.  The three margins share what the drift
    # leaves beyond the old bound
    ap64, ex64 = apP.astype(np.float64), exP.astype(np.float64)
    ok_self = np.abs(ap64 - ex64) <= 0.9 * e_old          # round 5's self-check would pass on them
    gap = exA - apA - e_old
    if gap <= 0:
        raise AssertionError(f"no drift beyond the old bound: exact - approx = {(exA - apA) / unit:.1f} units, eps_old 20")
    m = gap / 3.0
    f_idx = p_idx = None
    in_f = ok_self & (ap64 >= apA + 0.8 * m) & (ap64 <= apA + 1.3 * m)
    if in_f.any():
        f_idx = int(np.argmin(np.where(in_f, np.abs(ap64 - (apA + m)), np.inf)))
        U = ap64[f_idx]
        tgt = 0.5 * ((U + e_old) + exA)
        okp = ok_self & (ex64 > U + e_old) & (ex64 < exA) & (ap64 > U)
        if okp.any():
            p_idx = int(np.argmin(np.where(okp, np.abs(ex64 - tgt), np.inf)))
    if f_idx is None or p_idx is None:
        raise AssertionError("pool too thin for the construction")
    U = float(ap64[f_idx])
    margins = ((U - apA) / unit, (ex64[p_idx] - U - e_old) / unit, (exA - ex64[p_idx]) / unit)

    # the neighbourhood below A: pool rows whose approximate AND exact scores are below A's approx by at least 4 units
    below = np.where((ap64 < apA - 4 * unit) & (ex64 < apA - 4 * unit))[0][:4000]
    n_fill = n - copies - 2 * k - below.shape[0]
    KF = rng.integers(-127, 128, (n_fill, dim))  # (both signs: scores around zero, far below the construction)
    KF[np.arange(n_fill), rng.integers(0, dim, n_fill)] = 127
    K_all = np.concatenate([np.repeat(KP[f_idx][None, :], copies, 0), np.repeat(KP[p_idx][None, :], k, 0), KP[below], KF,
                            np.repeat(kA[None, :], k, 0)])
    rows = (K_all.astype(f32) * s).astype(f32)
    query = q.copy()
    if signed:  # the same products under a diagonal of signs: mixed-sign data, identical arithmetic
        D = rng.choice(np.array([-1.0, 1.0], f32), dim)
        rows *= D[None, :]
        query *= D
    info = dict(s=float(s), drift_units=drift / U24, eps_old=e_old, unit=unit, margins_units=margins, i8_rel=i8_rel, qrel=qrel,
                U=U, approx_A=apA, exact_A=exA, exact_P=float(ex64[p_idx]), copies=copies, k=k)
    return dict(rows=rows, query=query, expect_rows=np.arange(n - k, n), info=info)


def old_bound_outcome(rows, query, metric, k, T, e_old):
    """What a candidate pass certified with `e_old` returns on this corpus, by emulation: the T best approximate scores (ties: lower
    row first) re-scored exactly; (rows of its top-k, certified?)"""
    ap, _, _ = i8_approx(rows, query, metric)
    ex, _ = exact_scores(rows, query, metric)
    order = np.lexsort((np.arange(rows.shape[0]), -ap.astype(np.float64)))
    lst, outside = order[:T], float(ap[order[T]])
    top = lst[np.lexsort((lst, -ex[lst].astype(np.float64)))][:k]
    kth = float(ex[top[-1]])
    self_ok = bool((np.abs(ap[lst].astype(np.float64) - ex[lst]) <= e_old).all())
    return top, (kth > outside + e_old) and self_ok
