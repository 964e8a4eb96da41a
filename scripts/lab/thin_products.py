"""Wall time of thin products (one side 1..8, long k) — looking for serial paths. (lab script)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(0)
def t(name, fn):
    fn(); lib.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); lib.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:40s} {min(ts):9.3f} ms", flush=True)
for dt in (np.float32, np.float64, np.int64):
    for K in (4096, 1 << 16, 1 << 20, 1 << 22):
        x = nd.asarray(rng.standard_normal(K).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, K))
        y = nd.asarray(rng.standard_normal(K).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, K))
        t(f"{np.dtype(dt).name} dot 1d K={K}", lambda: nd.dot(x, y))
        t(f"{np.dtype(dt).name} (1,K)@(K,1) K={K}", lambda: nd.matmul(nd.reshape(x, (1, K)), nd.reshape(y, (K, 1))))
    K = 1 << 20
    A = nd.asarray(rng.standard_normal((8, K)).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, (8, K)))
    v = nd.asarray(rng.standard_normal(K).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, K))
    t(f"{np.dtype(dt).name} (8,K)@(K,) K=2^20", lambda: nd.matmul(A, v))
    t(f"{np.dtype(dt).name} (K,)@(K,8) K=2^20", lambda: nd.matmul(v, A.T))
    t(f"{np.dtype(dt).name} (8,K)@(K,8) K=2^20", lambda: nd.matmul(A, A.T))
    for (m, n) in ((16, 16), (64, 64), (9, 3), (1, 64), (64, 1), (12, 1), (1, 100)):
        Am = nd.asarray(rng.standard_normal((m, K)).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, (m, K)))
        Bm = nd.asarray(rng.standard_normal((K, n)).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, (K, n)))
        t(f"{np.dtype(dt).name} ({m},K)@(K,{n}) K=2^20", lambda: nd.matmul(Am, Bm))
    B = nd.asarray(rng.standard_normal((4096, 4096)).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, (4096, 4096)))
    w = nd.asarray(rng.standard_normal(4096).astype(dt) if dt is not np.int64 else rng.integers(-9, 9, 4096))
    t(f"{np.dtype(dt).name} (4096,4096)@(4096,)", lambda: nd.matmul(B, w))
    t(f"{np.dtype(dt).name} (4096,)@(4096,4096)", lambda: nd.matmul(w, B))
    t(f"{np.dtype(dt).name} outer (4096,1)@(1,4096)", lambda: nd.matmul(nd.reshape(w, (4096, 1)), nd.reshape(w, (1, 4096))))
