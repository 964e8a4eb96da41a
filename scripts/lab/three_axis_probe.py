"""Calls on 3-D / 4-D operands a normalisation or attention layer makes, one by one (wall time, GB/s of operands + result). (lab script)"""
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
    print(f"{name:60s} {min(ts)*1e3:9.3f} ms  {nbytes / min(ts) / 1e9:8.1f} GB/s", flush=True)
def arr(*shape, dt=np.float32): return nd.asarray(rng.standard_normal(shape).astype(dt))
B, R, C = 64, 512, 512
x = arr(B, R, C); n4 = x.size * 4
big = arr(B, R + 8, C + 8); v = big[:, 4:R + 4, 4:C + 4]
t("exp(sliced 3-D view)", lambda: nd.exp(v), 2 * n4)
t("copy(sliced 3-D view)", lambda: nd.copy(v), 2 * n4)
t("negative(x.transpose(1,0,2))", lambda: nd.negative(nd.transpose(x, (1, 0, 2))), 2 * n4)
t("negative(x.transpose(0,2,1))", lambda: nd.negative(nd.transpose(x, (0, 2, 1))), 2 * n4)
t("copy(x.transpose(0,2,1))", lambda: nd.copy(nd.transpose(x, (0, 2, 1))), 2 * n4)
t("copy(x.transpose(2,0,1))", lambda: nd.copy(nd.transpose(x, (2, 0, 1))), 2 * n4)
m = nd.greater(x, 0)
g = arr(B, 1, C); h = arr(1, R, 1)
t("where(mask, x, (B,1,C))", lambda: nd.where(m, x, g), 2 * n4 + x.size)
t("where(mask, (1,R,1), x)", lambda: nd.where(m, h, x), 2 * n4 + x.size)
t("where(mask, x, 0.0)", lambda: nd.where(m, x, 0.0), 2 * n4 + x.size)
t("greater(x, (B,1,C))", lambda: nd.greater(x, g), n4 + x.size)
o64 = nd.asarray(np.ones((B, 1, C)))
t("x * (B,1,C) float64*float32 mixed", lambda: nd.multiply(x, o64), n4 * 3)
for ax in (0, 1, 2, (0, 1), (0, 2), (1, 2)):
    t(f"sum(x, axis={ax}, keepdims=True)", lambda: nd.sum(x, axis=ax, keepdims=True), n4)
    t(f"max(x, axis={ax})", lambda: nd.max(x, axis=ax), n4)
t("mean(x, axis=1)", lambda: nd.mean(x, axis=1), n4)
t("std(x, axis=1)", lambda: nd.std(x, axis=1), n4)
t("std(x, axis=2)", lambda: nd.std(x, axis=2), n4)
t("argmax(x, axis=1)", lambda: nd.argmax(x, axis=1), n4)
x4 = arr(32, 64, 64, 128); n = x4.size * 4
c4, cw4, nh4, w2, g16, h16b = arr(1, 64, 1, 1), arr(1, 64, 1, 128), arr(32, 1, 64, 1), arr(512, 512), arr(B, 1, C, dt=np.float16), arr(1, R, 1, dt=np.float16)
t("(N,C,H,W) * (1,C,1,1)", lambda: nd.multiply(x4, c4), 2 * n)
t("(N,C,H,W) + (N,1,H,W)", lambda: nd.add(x4, nd.sum(x4, axis=1, keepdims=True)), 3 * n)
t("(N,C,H,W) - (N,C,1,1)", lambda: nd.subtract(x4, nd.mean(x4, axis=(2, 3), keepdims=True)), 3 * n)
t("(N,C,H,W) * (1,C,1,W)", lambda: nd.multiply(x4, cw4), 2 * n)
t("(N,C,H,W) * (N,1,H,1)", lambda: nd.multiply(x4, nh4), 2 * n)
t("sum((N,C,H,W), axis=(0,2,3))", lambda: nd.sum(x4, axis=(0, 2, 3)), n)
t("batched matmul (64,512,512)@(64,512,512)", lambda: nd.matmul(x, x), 3 * n4)
t("batched matmul x @ x.transpose(0,2,1)", lambda: nd.matmul(x, nd.transpose(x, (0, 2, 1))), 3 * n4)
t("batched matmul (64,512,512)@(512,512)", lambda: nd.matmul(x, w2), 2 * n4)
h16 = arr(B, R, C, dt=np.float16)
t("float16 (B,R,C) * (B,1,C)", lambda: nd.multiply(h16, g16), x.size * 4)
t("float16 (B,R,C) * (1,R,1)", lambda: nd.multiply(h16, h16b), x.size * 4)
nd.set_lazy(True)
gam, bet = arr(C), arr(C)
def ln():
    mu = nd.mean(x, axis=2, keepdims=True)
    d = nd.subtract(x, mu)
    va = nd.mean(nd.multiply(d, d), axis=2, keepdims=True)
    return nd.add(nd.multiply(nd.true_divide(d, nd.sqrt(nd.add(va, 1e-5))), gam), bet)
t("layer norm over the last axis, lazy", lambda: nd.materialize(ln()), 6 * n4)
t("lazy (x * (B,1,C) + (1,R,1)) ** 2", lambda: nd.materialize(nd.power(nd.add(nd.multiply(x, g), h), 2)), 2 * n4)
nd.set_lazy(False)
t("layer norm over the last axis, eager", ln, 12 * n4)
