"""Binary ops under the broadcast patterns of normalisation layers (wall time, GB/s of operands + result). (lab script)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(0)
def t(name, fn, nbytes):
    fn(); lib.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); lib.sync(); ts.append(time.perf_counter() - t0)
    print(f"{name:52s} {min(ts)*1e3:9.3f} ms  {nbytes / min(ts) / 1e9:8.1f} GB/s", flush=True)
def arr(*shape): return nd.asarray(rng.standard_normal(shape).astype(np.float32))
R, C = 4096, 4096
x = arr(R, C)
cases = [("(R,C) - (R,C)", x, arr(R, C)), ("(R,C) - (1,C) row vector", x, arr(1, C)), ("(R,C) - (C,)", x, arr(C)), ("(R,C) - (R,1) column", x, arr(R, 1)),
         ("(R,C) - 0-d", x, arr()), ("(R,C) - python scalar", x, 2.5), ("(R,C).T - (C,1)", x.T, arr(C, 1)), ("(R,C)[:, ::2] - (R,1)", x[:, ::2], arr(R, 1))]
for name, a, b in cases:
    n = a.size * 8 + (b.size * 4 if hasattr(b, "size") else 0)
    t("subtract " + name, lambda: nd.subtract(a, b), n)
    t("true_divide " + name, lambda: nd.true_divide(a, b), n)
B3 = arr(64, 512, 512)
for name, b in (("(B,R,C) * (B,1,C)", arr(64, 1, 512)), ("(B,R,C) * (1,R,1)", arr(1, 512, 1)), ("(B,R,C) * (B,R,1)", arr(64, 512, 1)), ("(B,R,C) * (B,1,1)", arr(64, 1, 1)),
                ("(B,R,C) * (1,1,C)", arr(1, 1, 512)), ("(B,R,C) * (R,C)", arr(512, 512))):
    t("multiply " + name, lambda: nd.multiply(B3, b), B3.size * 8 + b.size * 4)
B64 = nd.asarray(rng.standard_normal((64, 512, 512)))
b64 = nd.asarray(rng.standard_normal((64, 1, 512)))
t("multiply f64 (B,R,C) * (B,1,C)", lambda: nd.multiply(B64, b64), B64.size * 16)
Bv = arr(64, 520, 520)[:, 4:516, 4:516]
t("multiply (B,R,C) * sliced (B,R,C) view", lambda: nd.multiply(B3, Bv), B3.size * 12)
# the softmax chain, eager
def softmax():
    m = nd.max(x, axis=-1, keepdims=True)
    e = nd.exp(nd.subtract(x, m))
    return nd.true_divide(e, nd.sum(e, axis=-1, keepdims=True))
t("softmax over the last axis (5 calls)", softmax, x.size * 4 * 8)
nd.set_lazy(True)
t("softmax, lazy", lambda: nd.materialize(softmax()), x.size * 4 * 8)
nd.set_lazy(False)
