#!/usr/bin/env python3
"""A sweep for performance cliffs: every table function on a few-million-element operand in awkward layouts and dtypes; prints wall
time per call (after one warm-up call) and flags what is far off the bytes it moves. Not a benchmark — a smoke test for
serial / per-element-launch / host-loop paths. usage: perf_cliffs.py [elements, default 4e6]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

lib = _capi.load()
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
R = int(N ** 0.5)
rng = np.random.default_rng(0)
slow = []


def t(name, fn, nbytes=None):
    fn(); lib.sync()
    t0 = time.perf_counter()
    fn(); lib.sync()
    ms = (time.perf_counter() - t0) * 1e3
    flag = ""
    if ms > 20.0:
        flag = "   <-- SLOW"
        slow.append((name, ms))
    print(f"{name:58s} {ms:9.3f} ms{flag}", flush=True)


a2 = nd.asarray(rng.standard_normal((R, R)).astype(np.float32))
b2 = nd.asarray(rng.standard_normal((R, R)).astype(np.float32))
a3 = nd.asarray(rng.standard_normal((R // 8, 8, R)).astype(np.float32))
i2 = nd.asarray(rng.integers(-100, 100, (R, R)))
m2 = nd.asarray(rng.random((R, R)) > 0.5)
views = {"dense": a2, "T": a2.T, "cols::2": a2[:, ::2], "rows::2": a2[::2], "rev": a2[::-1, ::-1], "bcast row": nd.broadcast_to(a2[:1], (R, R)), "3d perm": nd.transpose(a3, (2, 0, 1))}
for vn, v in views.items():
    t(f"sin {vn}", lambda: nd.sin(v))
    t(f"add {vn} + dense", lambda: nd.add(v, b2[: v.shape[0], : v.shape[1]]) if v.ndim == 2 else nd.add(v, 1.0))
    t(f"sum all {vn}", lambda: nd.sum(v))
    t(f"sum axis0 {vn}", lambda: nd.sum(v, axis=0))
    t(f"sum axis-1 {vn}", lambda: nd.sum(v, axis=-1))
    t(f"max axis0 {vn}", lambda: nd.max(v, axis=0))
    t(f"argmax axis-1 {vn}", lambda: nd.argmax(v, axis=-1))
    t(f"copy {vn}", lambda: nd.copy(v, order="C"))
    t(f"astype f64 {vn}", lambda: v.astype(np.float64))
    t(f"where {vn}", lambda: nd.where(nd.greater(v, 0), v, 0.0))
    t(f"std axis0 {vn}", lambda: nd.std(v, axis=0))
for dt in (np.int8, np.uint8, np.int16, np.float16, np.uint32, np.uint64, np.int32, np.int64, np.float64, np.bool_):
    x = nd.asarray((rng.integers(0, 50, (R, R))).astype(dt)); y = nd.asarray((rng.integers(1, 50, (R, R))).astype(dt))
    t(f"add {np.dtype(dt).name}", lambda: nd.add(x, y))
    t(f"floor_divide {np.dtype(dt).name} T", lambda: nd.floor_divide(x.T, y))
    t(f"sum axis0 {np.dtype(dt).name}", lambda: nd.sum(x, axis=0))
    t(f"max {np.dtype(dt).name} T axis0", lambda: nd.max(x.T, axis=0))
    t(f"astype f32 {np.dtype(dt).name}", lambda: x.astype(np.float32))
    t(f"mean {np.dtype(dt).name}", lambda: nd.mean(x, axis=1))
    t(f"matmul 512 {np.dtype(dt).name}", lambda: nd.matmul(x[:512, :512], y[:512, :512]))
idx_rows = nd.asarray(rng.integers(0, R, R * 2)); idx_el = nd.asarray(rng.integers(0, R, N)); idx_el2 = nd.asarray(rng.integers(0, R, N))
t("gather rows a[idx]", lambda: a2[idx_rows])
t("gather cols a[:, idx]", lambda: a2[:, idx_rows])
t("gather elements a[i, j]", lambda: a2[idx_el, idx_el2])
t("gather mask a[m]", lambda: a2[m2])
t("take_along axis0", lambda: nd.take_along_axis(a2, nd.asarray(rng.integers(0, R, (R, R))), 0))
vals_rows = nd.asarray(rng.standard_normal((R * 2, R)).astype(np.float32)); vals_el = nd.asarray(rng.standard_normal(N).astype(np.float32))
t("index_add rows", lambda: nd.index_add(a2.copy(), idx_rows, vals_rows))
t("index_add cols", lambda: nd.index_add(a2.copy(), (slice(None), idx_rows), vals_rows.T))
t("index_add elements", lambda: nd.index_add(a2.copy(), (idx_el, idx_el2), vals_el))
t("index_add elements int", lambda: nd.index_add(i2.copy(), (idx_el, idx_el2), 1))
t("index_add histogram 10 bins", lambda: nd.index_add(nd.zeros((16,), np.float32), nd.asarray(rng.integers(0, 10, N)), vals_el))
t("setitem rows", lambda: a2.copy().__setitem__(idx_rows, vals_rows))
t("setitem elements", lambda: a2.copy().__setitem__((idx_el, idx_el2), vals_el))
t("setitem mask scalar", lambda: a2.copy().__setitem__(m2, 0.0))
t("put_along axis1", lambda: nd.put_along_axis(a2.copy(), nd.asarray(rng.integers(0, R, (R, 3))), 1.0, 1))
t("nonzero", lambda: nd.nonzero(m2))
t("argwhere", lambda: nd.argwhere(m2))
t("concatenate axis0", lambda: nd.concatenate([a2, b2], axis=0))
t("concatenate axis1", lambda: nd.concatenate([a2, b2], axis=1))
t("concatenate 64 pieces", lambda: nd.concatenate([a2[i * (R // 64):(i + 1) * (R // 64)] for i in range(64)], axis=0))
t("stack axis-1", lambda: nd.stack([a2, b2], axis=-1))
t("tile (2,2)", lambda: nd.tile(a2, (2, 2)))
t("repeat 3 axis1", lambda: nd.repeat(a2, 3, axis=1))
t("repeat 3 flat", lambda: nd.repeat(a2, 3))
t("split 8 + sum each", lambda: [nd.sum(p) for p in nd.split(a2[: R // 8 * 8], 8, axis=0)])
t("flip both", lambda: nd.copy(nd.flip(a2)))
t("isin 1e6 x 100", lambda: nd.isin(i2[:1000, :1000], nd.asarray(np.arange(100))))
t("unravel_index", lambda: nd.unravel_index(nd.asarray(rng.integers(0, N // 2, N // 4)), (R // 2, R)))
t("clip", lambda: nd.clip(a2, -0.5, 0.5))
t("power int", lambda: nd.power(i2, 3))
t("mod float", lambda: nd.mod(a2, 0.7))
t("tensordot axes=1", lambda: nd.tensordot(a2[:1024, :1024], b2[:1024, :1024], axes=1))
t("matmul 1000x1000 f64", lambda: nd.matmul(a2[:1000, :1000].astype(np.float64), b2[:1000, :1000].astype(np.float64)))
t("matmul ragged 1023x517x771", lambda: nd.matmul(a2[:1023, :517], b2[:517, :771]))
t("matmul batched 16x256x256", lambda: nd.matmul(nd.reshape(a2[:1024, :1024], (16, 256, 256)), nd.reshape(b2[:1024, :1024], (16, 256, 256))))
t("matmul T @ T", lambda: nd.matmul(a2[:1024, :1024].T, b2[:1024, :1024].T))
t("dot 1d", lambda: nd.dot(nd.ravel(a2), nd.ravel(b2)))
t("arange + reshape", lambda: nd.reshape(nd.arange(N), (R, -1)) if N % R == 0 else nd.arange(N))
t("full / zeros_like", lambda: (nd.full((R, R), 2.5), nd.zeros_like(a2)))
t("H2D + D2H 16 MB", lambda: nd.asarray(np.zeros(N, np.float32)).get())
# reductions of a 3-D operand over every axis set, in permuted layouts; narrow dtypes through the index kernels; many small calls
c3 = nd.asarray(rng.standard_normal((160, 160, 160)).astype(np.float32))
for vn, v in (("3d", c3), ("3d perm(2,0,1)", nd.transpose(c3, (2, 0, 1))), ("3d ::2", c3[::2, :, ::2])):
    for ax in (0, 1, 2, (0, 1), (0, 2), (1, 2)):
        t(f"sum {vn} axis={ax}", lambda: nd.sum(v, axis=ax))
    t(f"argmin {vn} axis=1", lambda: nd.argmin(v, axis=1))
    t(f"mean/std {vn} axis=(0,2)", lambda: (nd.mean(v, axis=(0, 2)), nd.std(v, axis=(0, 2))))
    t(f"any/all {vn}", lambda: (nd.any(nd.greater(v, 4.0), axis=1), nd.all(nd.less(v, 9.0))))
for dt in (np.int8, np.float16, np.uint16):
    x = nd.asarray(rng.integers(0, 50, (R, R)).astype(dt))
    vr = nd.asarray(rng.integers(0, 50, (R * 2, R)).astype(dt)); ve = nd.asarray(rng.integers(0, 50, N).astype(dt))
    t(f"gather rows {np.dtype(dt).name}", lambda: x[idx_rows])
    t(f"gather elements {np.dtype(dt).name}", lambda: x[idx_el, idx_el2])
    t(f"index_add rows {np.dtype(dt).name}", lambda: nd.index_add(x.copy(), idx_rows, vr))
    t(f"index_add elements {np.dtype(dt).name}", lambda: nd.index_add(x.copy(), (idx_el, idx_el2), ve))
    t(f"setitem elements {np.dtype(dt).name}", lambda: x.copy().__setitem__((idx_el, idx_el2), ve))
    t(f"sum axis0 / argmax {np.dtype(dt).name}", lambda: (nd.sum(x, axis=0), nd.argmax(x, axis=1)))
t("setitem mask <- values", lambda: a2.copy().__setitem__(m2, a2[m2]))
t("getitem a[i, :, j] on 3-d", lambda: c3[idx_rows % 160, :, idx_rows % 160])
t("isin 4e6 x 5000", lambda: nd.isin(i2, nd.asarray(np.arange(5000))))
small = [nd.asarray(np.arange(10.0)) for _ in range(8)]
t("1000 tiny adds (host overhead)", lambda: [nd.add(small[i % 8], small[(i + 1) % 8]) for i in range(1000)])
t("1000 tiny sums", lambda: [nd.sum(small[i % 8]) for i in range(1000)])
t("1000 tiny getitem int", lambda: [small[i % 8][3] for i in range(1000)])
t("100 tiny fancy getitem", lambda: [small[i % 8][nd.asarray(np.array([1, 2]))] for i in range(100)])
t("stack 200 small", lambda: nd.stack(small * 25))
# skinny geometries: very short rows / columns, unit extents, inner extent 1
for shp in ((N // 3, 3), (3, N // 3), (N // 2, 2), (1, N), (N, 1), (N // 64, 8, 8), (8, N // 64, 8), (8, 8, N // 64), (N // 5, 5, 1)):
    s_ = nd.asarray(rng.standard_normal(shp).astype(np.float32))
    nm = "x".join(str(v) for v in shp)
    for ax in range(len(shp)):
        t(f"sum {nm} axis={ax}", lambda: nd.sum(s_, axis=ax))
        t(f"argmax {nm} axis={ax}", lambda: nd.argmax(s_, axis=ax))
        t(f"max {nm} axis={ax} keepdims, subtract", lambda: nd.subtract(s_, nd.max(s_, axis=ax, keepdims=True)))
    t(f"sum {nm} all", lambda: nd.sum(s_))
    t(f"T copy {nm}", lambda: nd.copy(s_.T, order="C"))
    t(f"std {nm} axis=-1", lambda: nd.std(s_, axis=-1))
    t(f"add last-axis vector {nm}", lambda: nd.add(s_, nd.asarray(np.ones(shp[-1], np.float32))))
    t(f"multiply first-axis column {nm}", lambda: nd.multiply(s_, nd.asarray(np.ones(shp[:1] + (1,) * (len(shp) - 1), np.float32))))
    t(f"where {nm}", lambda: nd.where(nd.greater(s_, 0), s_, 0.0))
print(f"\n{len(slow)} calls over 20 ms:", slow)
