#!/usr/bin/env python3
"""perf_cliffs.py one level up: forward + backward of every differentiable op of the tape (this repo's mirror of the reference's
caller layer over the device table) on a few-million-element tensor; wall time per sweep, flags over 30 ms. usage: perf_cliffs_tape.py [elements]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402
from minidiff_amd.hip_backend import HipBackendTable  # noqa: E402
from minidiff_amd.tape import build_engine  # noqa: E402

lib = _capi.load()
md = build_engine(HipBackendTable, "dev")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
R = int(N ** 0.5)
rng = np.random.default_rng(0)
x = md.Tensor(rng.standard_normal((R, R)).astype(np.float32), allow_grad=True)
y = md.Tensor((rng.standard_normal((R, R)) + 3.0).astype(np.float32), allow_grad=True)
v = md.Tensor(rng.standard_normal((R,)).astype(np.float32), allow_grad=True)
idx = md.Tensor(rng.integers(0, R, R * 2))
aidx = md.Tensor(rng.integers(0, R, (R, R)))
slow = []


def t(name, f):
    def sweep():
        x.grad = y.grad = v.grad = None
        out = f()
        (md.sum(out) if out.shape != () else out).backward()
    try:
        sweep(); lib.sync()
    except Exception as e:      # (argument forms the reference's own backward rejects are preserved: not this script's business)
        print(f"{name:44s} raises {type(e).__name__}", flush=True)
        return
    t0 = time.perf_counter()
    sweep(); lib.sync()
    ms = (time.perf_counter() - t0) * 1e3
    flag = ""
    if ms > 30.0:
        flag = "   <-- SLOW"
        slow.append((name, ms))
    print(f"{name:44s} {ms:9.3f} ms{flag}", flush=True)


for name in ("sin", "cos", "tanh", "exp", "abs", "sqrt", "log", "square", "sign", "floor"):
    src = y if name in ("sqrt", "log") else x
    t(name, lambda: getattr(md, name)(src))
for name in ("add", "subtract", "multiply", "true_divide", "power", "mod", "floor_divide"):
    t(f"{name} full", lambda: getattr(md, name)(x, y))
    t(f"{name} row-broadcast", lambda: getattr(md, name)(x, md.abs(v) + 2.0))
    t(f"{name} scalar", lambda: getattr(md, name)(y, 2.5))
for name in ("sum", "mean", "max", "min", "prod", "std"):
    src = x if name != "prod" else md.tanh(x) * 0.001 + 1.0
    # (tuples: the reference's sum backward takes tuple(axis) and raises on an int; its max / min backward raises for axis=None — both preserved)
    for ax in ((None, (0,), (1,)) if name not in ("max", "min") else (0, 1)):
        t(f"{name} axis={ax}", lambda: getattr(md, name)(src, axis=ax) if name != "prod" else md.prod(md.tanh(x) * 0.001 + 1.0, axis=ax))
t("where", lambda: md.where(x > 0, x, y))
t("clip", lambda: md.clip(x, -0.5, 0.5))
t("matmul 1024", lambda: md.matmul(x[:1024, :1024], y[:1024, :1024]))
t("matmul x @ v", lambda: md.matmul(x, v))
t("matmul v @ x", lambda: md.matmul(v, x))
t("dot v . v", lambda: md.dot(v, v))
t("tensordot axes=2", lambda: md.tensordot(x, y, axes=2))
t("transpose + reshape", lambda: md.reshape(md.transpose(x), (-1,)) * 2.0)
t("swapaxes / expand / squeeze", lambda: md.squeeze(md.expand_dims(md.swapaxes(x, 0, 1), 0), axis=0) * 2.0)
t("flip", lambda: md.flip(x, axis=1) * 2.0)
t("getitem rows x[idx]", lambda: x[idx])
t("getitem cols x[:, idx]", lambda: x[:, idx])
t("getitem slice", lambda: x[10:-10:3, ::2])
t("getitem mask", lambda: x[x > 0])
t("take_along_axis", lambda: md.take_along_axis(x, aidx, 1))
t("concatenate + split", lambda: md.split(md.concatenate([x, y], axis=0), 2, axis=0)[1])
t("stack", lambda: md.stack([x, y], axis=0))
t("tile (2, 2)", lambda: md.tile(x, (2, 2)))
t("repeat 2 axis0", lambda: md.repeat(x, 2, axis=0))
t("broadcast_to", lambda: md.broadcast_to(v, (R, R)) * x)
t("astype f64", lambda: md.astype(x, md.float64))
t("copy / ravel / flatten", lambda: md.flatten(md.copy(x)) + md.ravel(y))
t("chain (sin x * y) ** 2", lambda: (md.sin(x) * y) ** 2)
print(f"\n{len(slow)} sweeps over 30 ms:", slow)
