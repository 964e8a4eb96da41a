#!/usr/bin/env python3
"""Differential fuzzing of forward+backward sweeps: a random expression graph over a few leaf
tensors (elementwise ops with broadcasting, matmul, reductions, indexing, where/clip, second
order) is run through the same tape on the device table and on the NumPy oracle table; outputs
and every leaf gradient must agree.   python tests/fuzz_tape.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minidiff_amd.hip_backend import HipBackendTable  # noqa: E402
from minidiff_amd.tape import build_engine  # noqa: E402
from oracle.numpy_table import NumpyOracleTable  # noqa: E402  (checker)

ENGINES = None


def engines():
    global ENGINES
    if ENGINES is None:
        ENGINES = (build_engine(HipBackendTable, "dev"), build_engine(NumpyOracleTable, "oracle"))
    return ENGINES


def build(md, rng, leaves_h, dt):
    """Replay the SAME random program (rng is re-seeded by the caller) on engine `md`."""
    leaves = [md.Tensor(h.copy(), allow_grad=True) for h in leaves_h]
    vals = list(leaves)
    n_ops = int(rng.integers(2, 9))
    for _ in range(n_ops):
        k = rng.choice(["un", "bin", "bin", "scalar", "red", "where", "clip", "mm", "idx", "reshape", "T"])
        a = vals[int(rng.integers(0, len(vals)))]
        if k == "un":
            f = str(rng.choice(["sin", "cos", "tanh", "exp", "absolute"]))
            x = a * 0.3 if f == "exp" else a
            vals.append(getattr(md, f)(x))
        elif k == "bin":
            b = vals[int(rng.integers(0, len(vals)))]
            f = str(rng.choice(["add", "subtract", "multiply", "true_divide"]))
            try:
                np.broadcast_shapes(a.shape, b.shape)
            except ValueError:
                continue
            if f == "true_divide":
                b = md.absolute(b) + 1.5
            vals.append(getattr(md, f)(a, b))
        elif k == "scalar":
            c = float(rng.choice([2.0, -0.5, 3.0]))
            vals.append(a * c if rng.random() < 0.5 else (a + c) ** 2)
        elif k == "red":
            if a.ndim == 0:
                continue
            f = str(rng.choice(["sum", "mean"]))
            ax = int(rng.integers(0, a.ndim))
            vals.append(getattr(md, f)(a, axis=ax, keepdims=bool(rng.random() < 0.5)))
        elif k == "where":
            vals.append(md.where(a > 0.1, a, a * 0.2))
        elif k == "clip":
            vals.append(md.clip(a, -0.7, 0.9))
        elif k == "mm":
            if a.ndim != 2:
                continue
            n = int(rng.integers(1, 7))
            w = md.Tensor(np.random.default_rng(int(rng.integers(0, 1 << 30))).standard_normal((a.shape[1], n)).astype(dt), allow_grad=True)
            leaves.append(w)
            vals.append(a @ w)
        elif k == "idx":
            if a.ndim == 0 or a.shape[0] == 0:
                continue
            idx = rng.integers(0, a.shape[0], (int(rng.integers(1, 6)),))
            vals.append(a[idx])
        elif k == "reshape":
            vals.append(md.reshape(a, (-1,)))
        elif k == "T":
            if a.ndim >= 2:
                vals.append(md.transpose(a))
    out = vals[-1]
    loss = md.sum(out * out) if rng.random() < 0.5 else md.sum(out)
    return leaves, out, loss


def run(md, seed, i, second):
    rng = np.random.default_rng([seed, i])
    dt = np.float32 if rng.random() < 0.5 else np.float64
    n_leaves = int(rng.integers(1, 4))
    base = tuple(int(x) for x in rng.integers(1, 6, int(rng.integers(1, 4))))
    shapes = [base]
    for _ in range(n_leaves - 1):
        s = list(base)
        for j in range(len(s)):
            if rng.random() < 0.3:
                s[j] = 1
        shapes.append(tuple(s[int(rng.integers(0, len(s))):]) if rng.random() < 0.5 else tuple(s))
    leaves_h = [rng.standard_normal(s).astype(dt) for s in shapes]
    leaves, out, loss = build(md, rng, leaves_h, dt)
    loss.backward(allow_higher_order=second)
    g1 = [None if t.grad is None else np.asarray(t.grad.as_numpy()).copy() for t in leaves]
    g2 = None
    if second and leaves[0].grad is not None:
        md.sum(leaves[0].grad * leaves[0].grad).backward()
        g2 = [None if t.grad is None else np.asarray(t.grad.as_numpy()).copy() for t in leaves]
    return np.asarray(out.as_numpy()), float(np.asarray(loss.as_numpy())), g1, g2, dt


def close(a, b, dt, what):
    assert (a is None) == (b is None), f"{what}: one side has no gradient"
    if a is None:
        return
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: {a.shape}/{a.dtype} vs {b.shape}/{b.dtype}"
    tol = 2e-4 if dt == np.float32 else 1e-10   # chains of up to 8 ops, second order included
    scale = max(float(np.abs(b).max()) if b.size else 0.0, 1e-6)
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if b.size else 0.0
    assert np.isfinite(a).all() == np.isfinite(b).all(), f"{what}: finiteness differs"
    if np.isfinite(b).all():
        assert err <= tol * scale, f"{what}: rel err {err / scale:.3e}"


def main(n=300, seed=0):
    dev, ora = engines()
    fails = 0
    for i in range(n):
        second = (i % 4 == 0)
        try:
            res = []
            for md in (dev, ora):   # forms the reference's gradient code rejects must be rejected alike
                try:
                    with np.errstate(all="ignore"):
                        res.append(run(md, seed, i, second))
                except (TypeError, ValueError, IndexError) as e:
                    res.append(e)
            if isinstance(res[0], Exception) or isinstance(res[1], Exception):
                assert type(res[0]) is type(res[1]), f"one engine raised: {res[0]!r} vs {res[1]!r}"
                continue
            (o_d, l_d, g_d, h_d, dt), (o_o, l_o, g_o, h_o, _) = res
            close(o_d, o_o, dt, "output")
            assert len(g_d) == len(g_o)
            for k, (a, b) in enumerate(zip(g_d, g_o)):
                close(a, b, dt, f"grad[{k}]")
            if second and h_o is not None:
                for k, (a, b) in enumerate(zip(h_d, h_o)):
                    close(a, b, dt, f"second-order grad[{k}]")
        except AssertionError as e:
            fails += 1
            print(f"FAIL case {i} (seed {seed}): {e}", flush=True)
        except Exception as e:
            fails += 1
            print(f"ERROR case {i} (seed {seed}): {type(e).__name__}: {e}", flush=True)
        if fails >= 15:
            break
    print(f"fuzz_tape: {n} cases, seed {seed}: {fails} failures", flush=True)
    return fails


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
