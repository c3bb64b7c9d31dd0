"""The cross-GPU merge on its own (what follows the all-gather of a sharded query): 8 shards' sorted candidate blocks, merged by
ott_merge_hits_device_grouped, wall per call (median of 30; includes the result's way to the host) with the rank merge
(default) and with the insertion merge (option merge_walk)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from otters_amd import VecStore
from otters_amd import _native as N
store = VecStore(8); store.add_vectors(np.ones((4, 8), np.float32))
L = N.lib(); rng = np.random.default_rng(1)
print("| shards | groups | list_len | k | rank merge us | insertion merge us |")
print("|---|---|---|---|---|---|")
for n_lists, n_groups, list_len, k in ((8, 1, 64, 10), (8, 1, 128, 100), (8, 256, 128, 100), (8, 1024, 128, 100), (8, 1, 512, 512), (64, 1, 64, 10), (64, 1, 128, 100)):
    lists = np.zeros((n_lists, n_groups, list_len), dtype=N.HIT_DTYPE)
    lists["index"] = np.uint64(0xFFFFFFFFFFFFFFFF); lists["score"] = np.float32(np.nan); lists["query"] = 0xFFFFFFFF
    cnt = min(k, list_len)
    sc = -np.sort(-rng.normal(0, 1, (n_lists, n_groups, cnt)).astype(np.float32), axis=2)
    lists["score"][:, :, :cnt] = sc
    lists["index"][:, :, :cnt] = rng.integers(0, 1 << 40, (n_lists, n_groups, cnt)).astype(np.uint64)
    lists["query"][:, :, :cnt] = np.arange(n_groups)[None, :, None]
    dev = torch.from_numpy(lists.view(np.uint8).reshape(-1).copy()).cuda()
    out = np.zeros(n_groups * min(k, n_lists * list_len), dtype=N.HIT_DTYPE)
    n_out = C.c_uint64(0); per = (C.c_uint64 * n_groups)()
    res = []
    for walk in (0, 1):
        store.set_option("merge_walk", walk)
        ts = []
        for it in range(40):
            t = time.perf_counter()
            N.check(L.ott_merge_hits_device_grouped(store._handle(), C.c_void_p(dev.data_ptr()), n_lists, n_groups, list_len, 1, k, N.ptr(out), C.byref(n_out), per))
            ts.append(time.perf_counter() - t)
        res.append(np.median(ts[10:]) * 1e6)
    print(f"| {n_lists} | {n_groups} | {list_len} | {k} | {res[0]:.1f} | {res[1]:.1f} |", flush=True)
