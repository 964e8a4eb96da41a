"""The forward+backward sweeps of BASELINE.json's configs, written against the
tape API so the same function runs on the HIP engine (product) and on the
NumPy-oracle engine (checker / CPU baseline). Shapes, seeds and sweep
definitions follow SURVEY.md §8(d) / BASELINE.md §3.

Every `make_*` returns (state, step) where `step()` runs ONE sweep and returns
the tensors whose gradients are the result; inputs are created once, before
any timing, and stay resident (HBM for the HIP engine).
"""
from __future__ import annotations

import numpy as np


def _finish(md, *tensors, outputs=()):
    """A sweep is complete when its gradients AND its forward output are IN MEMORY: in lazy mode this
    is what triggers the fused kernels and the deferred products (a no-op for eager engines). Without
    the output a lazily evaluated `C = A @ B; C.backward()` would never launch the forward product
    (nothing reads C) and the sweep would not be the workload it claims to be."""
    mat = getattr(md.backend, "_materialize_many", None)
    if mat is not None:
        mat([t.grad._data for t in tensors if t is not None and t.grad is not None] + [o._data for o in outputs])


def _randn(seed, shape, scale=1.0):
    a = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)
    if scale != 1.0:
        a *= np.float32(scale)
    return a


def make_cfg2(md, n=4096, seed=2, rank=0):
    """C = A @ B; C.backward()  — 3 GEMMs (NN, NT, TN). A is this rank's batch
    shard (rows are independent), B is the replicated parameter."""
    A = md.Tensor(_randn(seed + 1000 * rank, (n, n)), allow_grad=True)
    B = md.Tensor(_randn(seed + 7, (n, n)), allow_grad=True)

    def step():
        A.grad = None
        B.grad = None
        C = A @ B
        C.backward()
        _finish(md, A, B, outputs=(C,))
        return {"A": A, "B": B, "out": C}

    return {"A": A, "B": B, "params": [B], "flops": 3 * 2 * n ** 3, "rows": n}, step


def make_cfg3(md, n=100_000_000, seed=3):
    """sum((sin(x) * y) ** 2).backward() — 11 streaming kernels, 100*n bytes."""
    x = md.Tensor(_randn(seed, (n,)), allow_grad=True)
    y = md.Tensor(_randn(seed + 7, (n,)), allow_grad=True)

    def step():
        x.grad = None
        y.grad = None
        loss = md.sum((md.sin(x) * y) ** 2)
        loss.backward()
        _finish(md, x, y, outputs=(loss,))
        return {"x": x, "y": y, "out": loss}

    return {"x": x, "y": y, "params": [x, y], "bytes": 100 * n, "rows": n}, step


def make_cfg4(md, batch=8192, d_in=4096, d_out=4096, seed=4, rank=0, world=1):
    """sum(relu(X @ W + b)).backward(), relu := where(z > 0, z, 0); rank r owns
    rows [r*batch/world, (r+1)*batch/world) of X; W, b replicated."""
    from .dp import shard_rows
    shard = shard_rows(batch, rank, world)  # ValueError if the batch does not split evenly: no silently dropped rows
    rows = shard.stop - shard.start
    rng = np.random.default_rng(seed)
    Xfull = rng.standard_normal((batch, d_in), dtype=np.float32)
    Wh = rng.standard_normal((d_in, d_out), dtype=np.float32) / np.float32(64.0)
    bh = rng.standard_normal((d_out,), dtype=np.float32)
    X = md.Tensor(np.ascontiguousarray(Xfull[shard]))
    del Xfull
    W = md.Tensor(Wh, allow_grad=True)
    b = md.Tensor(bh, allow_grad=True)

    def step():
        W.grad = None
        b.grad = None
        z = X @ W + b
        loss = md.sum(md.where(z > 0, z, 0))
        loss.backward()
        _finish(md, W, b, outputs=(loss,))
        return {"W": W, "b": b, "out": loss}

    return {"X": X, "W": W, "b": b, "params": [W, b], "flops": 2 * 2 * rows * d_in * d_out, "rows": rows}, step


def make_cfg5(md, n=2048, seed=5):
    """Second order: C = A @ B; C.backward(allow_higher_order=True); A.grad.backward() — 5 GEMMs."""
    A = md.Tensor(_randn(seed, (n, n)), allow_grad=True)
    B = md.Tensor(_randn(seed + 7, (n, n)), allow_grad=True)

    def step():
        A.grad = None
        B.grad = None
        C = A @ B
        C.backward(allow_higher_order=True)
        A.grad.backward()
        _finish(md, A, B, outputs=(C,))
        return {"A": A, "B": B, "out": C}

    return {"A": A, "B": B, "params": [A, B], "flops": 5 * 2 * n ** 3, "rows": n}, step


def make_readme(md):
    """cfg1: the README example (reference README.md:13-36), int64 inputs."""
    x = md.Tensor([[0, 2, -2, 1], [-1, -1, -2, -2]], allow_grad=True)
    y = md.Tensor([[2, 3, 4, 5], [0, -1, -3, 2]], allow_grad=True)

    def step():
        f = 2 * y * md.sin(x) - x ** 2
        f.backward(allow_higher_order=True)
        first = (x.grad.as_numpy().copy(), y.grad.as_numpy().copy())
        x.grad.backward()
        second = (x.grad.as_numpy().copy(), y.grad.as_numpy().copy())
        return {"f": f.as_numpy(), "first": first, "second": second}

    return {"x": x, "y": y}, step


MAKERS = {"cfg2": make_cfg2, "cfg3": make_cfg3, "cfg4": make_cfg4, "cfg5": make_cfg5}
