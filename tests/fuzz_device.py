#!/usr/bin/env python3
"""Differential fuzzing of the device array library against NumPy: random shapes, dtypes, strided /
transposed / broadcast views and argument forms for the elementwise, reduction, arg-reduction,
indexing and matmul entry points. Integers / bools / indices must match bit for bit, floats within
a few ulp-scaled tolerances.   python tests/fuzz_device.py [n_cases] [seed] [--big] [--narrow]
--narrow adds the storage-only dtypes (int8/16, uint8/16/32/64, float16) to the dtype pool (native kernels since round 4).
Runs on whatever library the process binds (the GPU product by default; tests bind the CPU double)."""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import ndarray as nd  # noqa: E402

DTYPES = [np.float32, np.float64, np.int64, np.int32, np.bool_]
NARROW_DTYPES = [np.int8, np.int16, np.uint8, np.uint16, np.uint32, np.uint64, np.float16]


def rand_array(rng, shape, dt):
    if np.dtype(dt).kind == "f":
        a = (rng.standard_normal(shape) * rng.choice([0.1, 1.0, 30.0])).astype(dt)
        if a.size and rng.random() < 0.15:
            flat = a.reshape(-1)
            flat[rng.integers(0, flat.size, builtins_max(1, flat.size // 50))] = rng.choice([np.nan, np.inf, -np.inf, 0.0, -0.0])
        return a
    if np.dtype(dt).kind == "b":
        return rng.random(shape) < 0.5
    if np.dtype(dt).kind == "u":
        a = rng.integers(0, 19, shape).astype(dt)
        if a.size and rng.random() < 0.2:            # the top of the range too (wrap-around, uint64 >= 2**63)
            flat = a.reshape(-1)
            flat[rng.integers(0, flat.size, builtins_max(1, flat.size // 20))] = np.iinfo(dt).max - int(rng.integers(0, 3))
        return a
    return rng.integers(-9, 10, shape).astype(dt)


builtins_max = max


def rand_shape(rng, big):
    nd_ = int(rng.integers(0, 5))
    if big and rng.random() < 0.5:
        shape = [int(x) for x in rng.choice([1, 2, 3, 17, 64, 130, 257, 1024], nd_ or 1)][:3]
        while int(np.prod(shape)) > (1 << 22):       # bounded: <= 4M elements (32 MiB of f64) per operand
            shape[int(np.argmax(shape))] //= 4
        return tuple(shape)
    return tuple(int(x) for x in rng.integers(0 if rng.random() < 0.05 else 1, 9, nd_))


def rand_view(rng, h, d):
    """Apply the same random view ops to host and device arrays."""
    for _ in range(int(rng.integers(0, 3))):
        if h.ndim == 0:
            break
        k = rng.integers(0, 4)
        if k == 0 and h.ndim >= 2:
            perm = tuple(int(x) for x in rng.permutation(h.ndim))
            h, d = h.transpose(perm), nd.transpose(d, perm)
        elif k == 1:
            ax = int(rng.integers(0, h.ndim))
            n = h.shape[ax]
            if n >= 2:
                start, step = int(rng.integers(0, 2)), int(rng.choice([1, 2, -1]))
                sl = [slice(None)] * h.ndim
                sl[ax] = slice(None, None, step) if step < 0 else slice(start, None, step)
                h, d = h[tuple(sl)], d[tuple(sl)]
        elif k == 2 and h.ndim < 4:
            ax = int(rng.integers(0, h.ndim + 1))
            h, d = np.expand_dims(h, ax), nd.expand_dims(d, ax)
        elif k == 3 and h.ndim >= 2:
            a0, a1 = (int(x) for x in rng.choice(h.ndim, 2, replace=False))
            h, d = np.swapaxes(h, a0, a1), nd.swapaxes(d, a0, a1)
    return h, d


def broadcastable(rng, shape):
    s = list(shape)
    for i in range(len(s)):
        if rng.random() < 0.3:
            s[i] = 1
    cut = int(rng.integers(0, len(s) + 1))
    return tuple(s[cut:])


def close(got, exp, what):
    got = np.asarray(got)
    exp = np.asarray(exp)
    assert got.shape == exp.shape, f"{what}: shape {got.shape} vs {exp.shape}"
    assert got.dtype == exp.dtype, f"{what}: dtype {got.dtype} vs {exp.dtype}"
    if exp.dtype.kind in "biu":
        assert np.array_equal(got, exp), f"{what}: integer/bool mismatch"
        return
    tol = 3e-6 if exp.dtype == np.float32 else 1.5e-3 if exp.dtype == np.float16 else 1e-13
    with np.errstate(all="ignore"):
        same_special = np.array_equal(np.isnan(got), np.isnan(exp)) and np.array_equal(np.isinf(got), np.isinf(exp))
        assert same_special, f"{what}: nan/inf pattern differs"
        fin = np.isfinite(exp)
        if fin.any():
            err = np.abs(got[fin].astype(np.float64) - exp[fin].astype(np.float64))
            bound = tol * np.maximum(np.abs(exp[fin]).astype(np.float64), 1e-30) + tol * (1e-2 if exp.dtype != np.float16 else 1e-4)
            assert (err <= bound).all(), f"{what}: max rel err {(err / np.maximum(np.abs(exp[fin]), 1e-30)).max():.3e}"
        inf = np.isinf(exp)
        assert np.array_equal(got[inf], exp[inf]), f"{what}: inf sign"


UNARY = ["absolute", "negative", "sign", "ceil", "floor", "sin", "cos", "tanh", "exp", "sqrt", "logical_not", "copy"]
BINARY = ["add", "subtract", "multiply", "true_divide", "maximum", "minimum", "less", "greater_equal", "equal", "not_equal",
          "logical_and", "logical_or", "floor_divide", "mod", "power"]
REDUCE = ["sum", "prod", "max", "min", "any", "all", "mean"]


def one_case(rng, big):
    kind = rng.choice(["unary", "binary", "where", "reduce", "arg", "gather", "scatter", "matmul", "astype",
                       "inplace", "concat", "along", "misc", "kwargs"])
    dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
    shape = rand_shape(rng, big)
    h = rand_array(rng, shape, dt)
    d = nd.asarray(h)
    h, d = rand_view(rng, h, d)
    with np.errstate(all="ignore"):
        if kind == "unary":
            name = str(rng.choice(UNARY))
            if name in ("sin", "cos", "tanh", "exp", "sqrt") and h.dtype.kind == "f":
                h = np.clip(np.nan_to_num(h, nan=0.5, posinf=3.0, neginf=-3.0), -20, 20).astype(h.dtype)
                d = nd.asarray(h)
            if name == "negative" and h.dtype == np.bool_:
                return
            if name == "sign" and h.dtype == np.bool_:
                return
            close(getattr(nd, name)(d), getattr(np, name)(h), f"{name}{h.shape}{h.dtype}")
        elif kind == "binary":
            name = str(rng.choice(BINARY))
            dt2 = DTYPES[int(rng.integers(0, len(DTYPES)))]
            if rng.random() < 0.25:
                h2 = d2 = rng.choice([2, -3, 0.5, 2.0, True]).item()
            else:
                h2 = rand_array(rng, broadcastable(rng, h.shape), dt2)
                d2 = nd.asarray(h2)
            if name == "power":
                # keep magnitudes representable; integer bases get small non-negative or (to check the
                # error path) possibly negative integer exponents
                if np.asarray(h2).dtype.kind == "f":
                    h2 = np.clip(np.nan_to_num(np.asarray(h2), nan=1.0, posinf=2.0, neginf=-2.0), -3, 3).astype(np.asarray(h2).dtype)
                    h2 = h2 if np.ndim(h2) else h2.item()
                    d2 = nd.asarray(h2) if isinstance(h2, np.ndarray) else h2
                if np.asarray(h).dtype.kind == "f":
                    h = np.clip(np.nan_to_num(h, nan=1.0, posinf=2.0, neginf=-2.0), -4, 4).astype(h.dtype)
                    d = nd.asarray(h)
            if name in ("subtract",) and (np.asarray(h).dtype == np.bool_ and np.asarray(h2).dtype == np.bool_):
                return
            if rng.random() < 0.5:
                h, h2, d, d2 = h2, h, d2, d
            if not isinstance(d, nd.DeviceArray) and not isinstance(d2, nd.DeviceArray):
                return
            try:
                exp = getattr(np, name)(h, h2)
            except (ValueError, TypeError) as e:   # e.g. integers to negative integer powers
                try:
                    getattr(nd, name)(d, d2)
                except type(e):
                    return
                raise AssertionError(f"{name}: NumPy raised {type(e).__name__}, the device did not")
            if exp.dtype.kind == "f" and name == "power":   # powf/pow: a few ulp on the device's libm
                got = np.asarray(getattr(nd, name)(d, d2))
                assert got.shape == exp.shape and got.dtype == exp.dtype
                fin = np.isfinite(exp) & np.isfinite(got)
                assert np.array_equal(np.isnan(got), np.isnan(exp)), f"power nan pattern {np.shape(h)} {np.shape(h2)}"
                tol = 2e-5 if exp.dtype == np.float32 else 1e-12
                assert (np.abs(got[fin] - exp[fin]) <= tol * np.maximum(np.abs(exp[fin]), 1e-30)).all(), f"power values {np.asarray(h).dtype} {np.asarray(h2).dtype}"
                return
            close(getattr(nd, name)(d, d2), exp, f"{name} {np.shape(h)}{np.asarray(h).dtype} {np.shape(h2)}{np.asarray(h2).dtype}")
        elif kind == "where":
            c = rng.random(h.shape) < 0.5
            b = rand_array(rng, broadcastable(rng, h.shape), dt)
            close(nd.where(nd.asarray(c), d, nd.asarray(b)), np.where(c, h, b), f"where {h.shape}{dt}")
        elif kind == "reduce":
            name = str(rng.choice(REDUCE))
            if h.ndim == 0 or h.size == 0:
                return
            if rng.random() < 0.3:
                axis = None
            else:
                k = int(rng.integers(1, h.ndim + 1))
                axis = tuple(int(x) for x in rng.choice(h.ndim, k, replace=False))
                if len(axis) == 1 and rng.random() < 0.5:
                    axis = axis[0] - (h.ndim if rng.random() < 0.3 else 0)
            keep = bool(rng.random() < 0.4)
            if h.dtype.kind == "f":
                h = np.nan_to_num(h, nan=0.25, posinf=4.0, neginf=-4.0).astype(h.dtype)
                if name == "prod":
                    h = (np.sign(h) * (0.999 + 0.002 * np.abs(np.tanh(h)))).astype(h.dtype)   # keeps the product in range
                d = nd.asarray(h)
            exp = getattr(np, name)(h, axis=axis, keepdims=keep)
            got = getattr(nd, name)(d, axis=axis, keepdims=keep)
            if exp.dtype.kind == "f" and name in ("sum", "mean", "prod"):
                got, exp = np.asarray(got), np.asarray(exp)
                assert got.shape == exp.shape and got.dtype == exp.dtype, f"{name} {h.shape} axis={axis}: {got.shape}/{got.dtype} vs {exp.shape}/{exp.dtype}"
                scale = np.abs(h.astype(np.float64)).sum() + 1.0 if name != "prod" else np.abs(exp).max() + 1.0
                tol = 2e-6 if h.dtype == np.float32 else 2e-3 if h.dtype == np.float16 else 1e-13
                # a product's rounding error grows with the number of factors and depends on their order (tree here, pairwise
                # blocks in NumPy): ~sqrt(n) ulp typically — 1e6 factors once missed a flat 50-ulp bound (seed 101 --big)
                n_red = max(h.size // max(exp.size, 1), 1)
                slack = 50 * max(1.0, (n_red / 4096.0) ** 0.5) if name == "prod" else 1
                if name == "prod":
                    # .. and when many factors are IDENTICAL (a saturated tanh makes them all 1.001) the two halves of a tree node
                    # carry the same rounding error, which then doubles per level instead of random-walking: the a-priori bound
                    # (n - 1) u is attained in order of magnitude (3e-4 on 33,410 float32 factors, seed 71 --big case 3138); NumPy's
                    # sequential product does not meet that case. The bound below is that a-priori bound, n * eps.
                    slack = max(slack, n_red * float(np.finfo(h.dtype).eps) / tol)
                fin = np.isfinite(exp)    # (a product of 1e6 factors may overflow on both sides: non-finite entries must simply agree)
                assert np.array_equal(got[~fin], exp[~fin], equal_nan=True), f"{name} {h.shape}{h.dtype} axis={axis}: non-finite entries differ"
                if not fin.all():
                    scale = np.abs(exp[fin]).max() + 1.0 if (name == "prod" and fin.any()) else scale
                if fin.any():
                    assert np.abs(got[fin].astype(np.float64) - exp[fin].astype(np.float64)).max() <= tol * scale * slack, f"{name} {h.shape}{h.dtype} axis={axis}"
            else:
                close(got, exp, f"{name} {h.shape}{h.dtype} axis={axis} keep={keep}")
        elif kind == "arg":
            if h.size == 0:
                return
            axis = None if (h.ndim == 0 or rng.random() < 0.3) else int(rng.integers(-h.ndim, h.ndim))
            name = str(rng.choice(["argmax", "argmin"]))
            close(getattr(nd, name)(d, axis=axis), getattr(np, name)(h, axis=axis), f"{name} {h.shape}{h.dtype} axis={axis}")
        elif kind == "gather":
            if h.ndim == 0 or h.size == 0:
                return
            ax = int(rng.integers(0, h.ndim))
            ish = rand_shape(rng, False)[:2]
            idx = rng.integers(-h.shape[ax], h.shape[ax], ish)
            key = [slice(None)] * h.ndim
            key[ax] = idx
            dkey = list(key)
            dkey[ax] = nd.asarray(idx)
            close(d[tuple(dkey)], h[tuple(key)], f"gather {h.shape}{h.dtype} ax={ax} idx{idx.shape}")
        elif kind == "scatter":
            if h.ndim == 0 or h.size == 0 or h.dtype == np.bool_:
                return
            h = np.ascontiguousarray(h)
            d = nd.asarray(h)
            ax = int(rng.integers(0, h.ndim))
            n = int(rng.integers(1, 12 if not big else 3000))
            # (a value of at most 2e7 elements: 2608 repeats of a 2-million-element slab once drew 5.6e9 positions — 7.5 minutes in NumPy alone)
            n = min(n, max(1, 20_000_000 // max(h.size // h.shape[ax], 1)))
            idx = rng.integers(-h.shape[ax], h.shape[ax], (n,))
            vshape = h.shape[:ax] + (n,) + h.shape[ax + 1:]
            vals = rand_array(rng, vshape, h.dtype)
            if h.dtype.kind == "f":
                vals = np.nan_to_num(vals, nan=1.0, posinf=2.0, neginf=-2.0).astype(h.dtype)
                h = np.nan_to_num(h, nan=1.0, posinf=2.0, neginf=-2.0).astype(h.dtype)
                d = nd.asarray(h)
            key = tuple([slice(None)] * ax + [idx])
            dkey = tuple([slice(None)] * ax + [nd.asarray(idx)])
            exp = h.copy()
            if rng.random() < 0.5:
                np.add.at(exp, key, vals)
                nd.index_add(d, dkey, nd.asarray(vals))
            else:
                exp[key] = vals
                d[dkey] = nd.asarray(vals)
            got = np.asarray(d)
            assert np.array_equal(got, exp), f"scatter {h.shape}{h.dtype} ax={ax} n={n}"
        elif kind == "matmul":
            fdt = np.float32 if rng.random() < 0.5 else np.float64
            lim = 300 if big else 40
            M, K, N = (int(x) for x in rng.integers(1, lim, 3))
            if big and rng.random() < 0.35:
                # the shape classes with kernels of their own: a thin side (skinny.hip), sizes a few past a multiple of 256 (peeled
                # products), whole direct-to-LDS tiles
                fam = rng.choice(["thin_n", "thin_m", "peel", "tiles"])
                if fam == "thin_n":
                    M, K, N = int(rng.choice([512, 1000, 1024, 2052, 4096])), int(rng.choice([64, 260, 512, 1024, 2048])), int(rng.integers(1, 9))
                elif fam == "thin_m":
                    M, K, N = int(rng.integers(1, 9)), int(rng.choice([64, 260, 512, 1024, 2048])), int(rng.choice([512, 1000, 1024, 2052, 4096]))
                elif fam == "peel":
                    M, K, N = int(rng.choice([256, 512, 768])) + int(rng.integers(0, 10)), 32 * int(rng.integers(1, 9)), int(rng.choice([256, 512, 768])) + int(rng.integers(0, 10))
                else:
                    M, K, N = (int(rng.choice([128, 256, 384, 512])) for _ in range(3))
            a = rng.standard_normal((M, K)).astype(fdt)
            b = rng.standard_normal((K, N)).astype(fdt)
            A, B = nd.asarray(a), nd.asarray(b)
            if rng.random() < 0.5:
                A = nd.asarray(np.ascontiguousarray(a.T)).T
            if rng.random() < 0.5:
                B = nd.asarray(np.ascontiguousarray(b.T)).T
            got = np.asarray(nd.matmul(A, B))
            ref = a.astype(np.float64) @ b.astype(np.float64)
            tol = 3e-6 if fdt == np.float32 else 1e-13
            assert got.dtype == fdt and np.abs(got - ref).max() <= tol * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)).max() + 1e-30, f"matmul {M}x{K}x{N} {fdt}"
        elif kind == "inplace":
            # views of one block written in place: the host copy must change the same way
            h = np.array(h, order="C")
            d = nd.asarray(h)
            hv, dv = rand_view(rng, h, d)
            k = rng.integers(0, 4)
            other = rand_array(rng, broadcastable(rng, hv.shape), hv.dtype)
            if hv.dtype.kind == "f":
                other = np.nan_to_num(other, nan=1.0, posinf=2.0, neginf=-2.0).astype(hv.dtype)
            if k == 0 and hv.dtype != np.bool_:
                hv += other
                dv += nd.asarray(other)
            elif k == 1 and hv.dtype != np.bool_:
                hv *= 2
                dv *= 2
            elif k == 2:
                hv[...] = other
                dv[...] = nd.asarray(other)
            elif hv.ndim >= 1 and hv.shape[0] >= 2:
                # operand overlapping the destination: NumPy computes as if from a copy
                if hv.dtype != np.bool_ and rng.random() < 0.5:
                    hv[1:] += hv[:-1]
                    dv[1:] += dv[:-1]
                else:
                    hv[1:] = hv[:-1]
                    dv[1:] = dv[:-1]
            if hv.ndim == 2 and hv.shape[0] == hv.shape[1] and hv.dtype != np.bool_ and rng.random() < 0.5:
                hv += hv.T
                dv += dv.T
            close(d, h, f"inplace k={k} {hv.shape}{hv.dtype} strides={hv.strides}")
        elif kind == "concat":
            if h.ndim == 0:
                return
            ax = int(rng.integers(0, h.ndim))
            parts_h, parts_d = [h], [d]
            for _ in range(int(rng.integers(1, 3))):
                sh = list(h.shape)
                sh[ax] = int(rng.integers(0, 4))
                p = rand_array(rng, tuple(sh), h.dtype)
                parts_h.append(p)
                parts_d.append(nd.asarray(p))
            close(nd.concatenate(parts_d, axis=ax), np.concatenate(parts_h, axis=ax), f"concatenate {h.shape} ax={ax}")
            sax = int(rng.integers(0, h.ndim + 1))
            close(nd.stack([d, d], axis=sax), np.stack([h, h], axis=sax), f"stack {h.shape} ax={sax}")
        elif kind == "along":
            if h.ndim == 0 or h.size == 0 or h.dtype == np.bool_:
                return
            ax = int(rng.integers(0, h.ndim))
            ish = list(h.shape)
            ish[ax] = int(rng.integers(1, 4))
            idx = rng.integers(0, h.shape[ax], tuple(ish))
            close(nd.take_along_axis(d, nd.asarray(idx), axis=ax), np.take_along_axis(h, idx, axis=ax), f"take_along_axis {h.shape} ax={ax}")
            am = np.argmax(np.nan_to_num(h), axis=ax, keepdims=True) if h.dtype.kind == "f" else np.argmax(h, axis=ax, keepdims=True)
            hz = np.zeros(h.shape, dtype=h.dtype)
            dz = nd.zeros(h.shape, dtype=h.dtype)
            vals = rand_array(rng, am.shape, h.dtype)
            np.put_along_axis(hz, am, vals, axis=ax)
            nd.put_along_axis(dz, nd.asarray(am), nd.asarray(vals), axis=ax)
            close(dz, hz, f"put_along_axis {h.shape} ax={ax}")
        elif kind == "misc":
            k = rng.integers(0, 5)
            if k == 0 and h.ndim:
                ax = None if rng.random() < 0.3 else int(rng.integers(0, h.ndim))
                close(nd.flip(d, axis=ax), np.flip(h, axis=ax), f"flip {h.shape} ax={ax}")
            elif k == 1:
                reps = tuple(int(x) for x in rng.integers(1, 3, int(rng.integers(1, 4))))
                close(nd.tile(d, reps), np.tile(h, reps), f"tile {h.shape} reps={reps}")
            elif k == 2 and h.ndim:
                ax = int(rng.integers(0, h.ndim))
                close(nd.repeat(d, 2, axis=ax), np.repeat(h, 2, axis=ax), f"repeat {h.shape} ax={ax}")
            elif k == 3 and h.dtype.kind == "f":
                close(nd.clip(d, -0.5, 0.75), np.clip(h, -0.5, 0.75), f"clip {h.shape}")
            elif k == 4 and h.dtype.kind == "f" and h.ndim and h.size:
                hh = np.nan_to_num(h, nan=0.25, posinf=4.0, neginf=-4.0).astype(h.dtype)
                ax = int(rng.integers(0, h.ndim))
                got, exp = np.asarray(nd.std(nd.asarray(hh), axis=ax)), np.std(hh, axis=ax)
                assert got.shape == exp.shape and got.dtype == exp.dtype
                # (float16: NumPy rounds every step of its five to half; the device computes them in float32 and rounds once)
                tol = 2e-5 if h.dtype == np.float32 else 4e-3 if h.dtype == np.float16 else 1e-11
                g64, e64 = got.astype(np.float64), exp.astype(np.float64)
                if h.dtype == np.float16:
                    # (.. so NumPy's half sum of squares overflows to inf from |x| ~ 30 on a 64-element axis where the device's
                    # float32 sum does not: compared where NumPy stayed finite)
                    fin = np.isfinite(e64)
                    g64, e64 = g64[fin], e64[fin]
                if e64.size:
                    assert np.abs(g64 - e64).max() <= tol * (float(np.abs(hh).max()) + 1.0), f"std {h.shape} ax={ax}"
        elif kind == "kwargs":
            # keyword forms of the table's functions (out=, dtype=, order=, invert=, scalar-boolean keys ..) on the drawn view
            k = int(rng.integers(0, 7))
            if k == 0 and h.dtype != np.bool_:          # reduction with dtype= and out=
                name = str(rng.choice(["sum", "prod"]))
                axis = None if h.ndim == 0 else int(rng.integers(0, h.ndim))
                rdt = np.float64 if h.dtype.kind == "f" else np.int64
                hh = h if name == "sum" or h.dtype.kind != "f" else (np.sign(h) * (0.999 + 0.002 * np.abs(np.tanh(np.nan_to_num(h.astype(np.float64)))))).astype(h.dtype)
                hh = np.nan_to_num(hh, nan=0.25, posinf=2.0, neginf=-2.0).astype(h.dtype) if h.dtype.kind == "f" else hh
                exp = getattr(np, name)(hh, axis=axis, dtype=rdt)
                o = nd.zeros(exp.shape, exp.dtype)
                got = getattr(nd, name)(nd.asarray(hh), axis=axis, dtype=rdt, out=o)
                assert got is o
                if rdt is np.int64:
                    assert np.array_equal(np.asarray(got), exp), f"{name}(dtype, out) {h.shape}{h.dtype}"
                else:
                    assert np.allclose(np.asarray(got), exp, rtol=1e-9, atol=1e-9 * (np.abs(hh.astype(np.float64)).sum() + 1)), f"{name}(dtype, out) {h.shape}{h.dtype}"
            elif k == 1:                                 # ufunc with out= of a wider dtype
                odt = np.float64 if h.dtype.kind in "fiub" else h.dtype
                exp = np.zeros(h.shape, odt); np.add(h, h, out=exp)
                o = nd.zeros(h.shape, odt)
                assert nd.add(d, d, out=o) is o
                close(o, exp, f"add(out={np.dtype(odt).name}) {h.shape}{h.dtype}")
            elif k == 2:                                 # memory orders
                order = str(rng.choice(["C", "F", "A", "K"]))
                close(nd.ravel(d, order=order), np.ravel(h, order=order), f"ravel {order} {h.shape}{h.dtype}")
                c, e = nd.copy(d, order=order), np.copy(h, order=order)
                close(c, e, f"copy {order} {h.shape}{h.dtype}")
                assert c.is_c_contiguous == e.flags.c_contiguous and c.T.is_c_contiguous == e.flags.f_contiguous, f"copy {order} layout {h.shape}"
                close(nd.flatten(d, order=order), h.flatten(order=order), f"flatten {order} {h.shape}{h.dtype}")
            elif k == 3:                                 # scalar booleans, None and Ellipsis in keys
                parts = [bool(rng.integers(0, 2)), None, Ellipsis, slice(None, None, int(rng.choice([1, 2, -1])))]
                key = tuple(parts[int(i)] for i in rng.permutation(4)[: int(rng.integers(1, 4))])
                if sum(1 for q in key if isinstance(q, slice)) > h.ndim:
                    return
                close(d[key], h[key], f"getitem {key!r} {h.shape}{h.dtype}")
            elif k == 4 and h.dtype.kind in "iu" and h.size:
                te = rng.integers(-3, 4, int(rng.integers(0, 6))).astype(h.dtype)
                inv = bool(rng.integers(0, 2))
                close(nd.isin(d, nd.asarray(te), invert=inv), np.isin(h, te, invert=inv), f"isin invert={inv} {h.shape}{h.dtype}")
            elif k == 5 and h.size and h.ndim:
                order = str(rng.choice(["C", "F"]))
                flat = rng.integers(0, h.size, (5,))
                for g, e in zip(nd.unravel_index(nd.asarray(flat), h.shape, order=order), np.unravel_index(flat, h.shape, order=order)):
                    close(g, e, f"unravel_index {order} {h.shape}")
            elif k == 6 and h.dtype != np.bool_:         # mean with an integer dtype=, std with ddof
                if h.ndim == 0 or h.size == 0:
                    return
                axis = int(rng.integers(0, h.ndim))
                if h.dtype.kind in "iu":
                    close(nd.mean(d, axis=axis, dtype=np.int64), np.mean(h, axis=axis, dtype=np.int64), f"mean(dtype=int64) {h.shape}{h.dtype}")
        elif kind == "astype":
            to = DTYPES[int(rng.integers(0, len(DTYPES)))]
            if h.dtype.kind == "f" and np.dtype(to).kind in "iu":
                h = np.nan_to_num(h, nan=1.0, posinf=2.0, neginf=-2.0).astype(h.dtype)
                d = nd.asarray(h)
            close(nd.astype(d, to), h.astype(to), f"astype {h.shape} {h.dtype}->{np.dtype(to)}")


def main(n=2000, seed=0, big=False, narrow=False):
    if narrow and NARROW_DTYPES[0] not in DTYPES:
        DTYPES.extend(NARROW_DTYPES)
    fails = 0
    for i in range(n):
        rng = np.random.default_rng([seed, i])
        try:
            one_case(rng, big)
        except AssertionError as e:
            fails += 1
            print(f"FAIL case {i} (seed {seed}): {e}", flush=True)
        except Exception as e:  # the two sides must also agree on raising
            fails += 1
            print(f"ERROR case {i} (seed {seed}): {type(e).__name__}: {e}", flush=True)
            if fails <= 3:
                traceback.print_exc()
        if fails >= 25:
            break
        if (i + 1) % 500 == 0:
            print(f"  .. {i + 1} cases, {fails} failures", flush=True)     # (a long run must keep writing: a silent GPU job is taken for hung)
    print(f"fuzz: {n} cases, seed {seed}, big={big}, narrow={narrow}: {fails} failures", flush=True)
    return fails


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sys.exit(1 if main(int(args[0]) if args else 2000, int(args[1]) if len(args) > 1 else 0, "--big" in sys.argv, "--narrow" in sys.argv) else 0)
