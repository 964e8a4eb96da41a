#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by running the REAL reference
(/root/reference, NumPy backend) in this container.

    python tests/golden/make_golden.py

The reference is imported unmodified (sys.argv is reset first because
minidiff/backend/__init__.py:17 parses it at import). Only DATA is written:
inputs, forward outputs, gradients, exception type names and ordered
backend-call traces. Nothing of the reference's source is stored.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MINIDIFF_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
sys.argv = [sys.argv[0], "--backend", "_trace_backend"]

import numpy as np  # noqa: E402

import minidiff as md  # noqa: E402
import minidiff.backend as mdb  # noqa: E402
import _trace_backend as tb  # noqa: E402

assert mdb.sin.__name__ == "sin" and "traced" in repr(mdb.sin), "tracing backend was not selected"
import cases  # noqa: E402


def to_np(t):
    return np.asarray(t._data)


def run_case(case, dtype):
    ins = []
    for a in case["inputs"]:
        if a.dtype.kind == "f" and dtype.startswith("float"):
            a = a.astype(dtype)
        elif a.dtype.kind in "iu" and case["name"].startswith("narrow_int") and dtype != "asis" and not (case["name"] == "narrow_int_getitem" and a.ndim == 1):
            a = a.astype(dtype)   # (storage-only integer dtypes; the index array of the gather case stays int64)
        ins.append(a)
    rec = {"inputs": ins}
    try:
        tensors = [md.Tensor(a.copy(), allow_grad=g) for a, g in zip(ins, case["grads"])]
        out = case["fn"](md, *tensors)
        rec["forward"] = to_np(out).copy()
    except Exception as e:  # the reference's own behaviour, recorded as data
        rec["forward_raises"] = type(e).__name__
        return rec
    if not any(case["grads"]) or not out.allow_grad:
        return rec
    try:
        loss = cases.loss_of(md, out)
        rec["loss"] = to_np(loss).copy()
        loss.backward()
        rec["grads"] = [None if t.grad is None else to_np(t.grad).copy() for t in tensors]
    except Exception as e:
        rec["backward_raises"] = type(e).__name__
    return rec


def gen_ops():
    arrays, meta = {}, {}
    for case in cases.CASES:
        for dtype in case["dtypes"]:
            key = f"{case['name']}@{dtype}"
            rec = run_case(case, dtype)
            m = {"n_inputs": len(rec["inputs"])}
            for i, a in enumerate(rec["inputs"]):
                arrays[f"{key}/in{i}"] = a
            if "forward_raises" in rec:
                m["forward_raises"] = rec["forward_raises"]
            else:
                arrays[f"{key}/forward"] = rec["forward"]
            if "backward_raises" in rec:
                m["backward_raises"] = rec["backward_raises"]
            if "loss" in rec:
                arrays[f"{key}/loss"] = rec["loss"]
            if "grads" in rec:
                m["grad_present"] = [g is not None for g in rec["grads"]]
                for i, g in enumerate(rec["grads"]):
                    if g is not None:
                        arrays[f"{key}/grad{i}"] = g
            meta[key] = m
    return arrays, meta


def traced(fn):
    tb.LOG.clear()
    tb.ENABLED[0] = True
    try:
        out = fn()
    finally:
        tb.ENABLED[0] = False
    return out, [list(x) for x in tb.LOG]


def f32(seed, shape, scale=1.0):
    a = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)
    return a * np.float32(scale) if scale != 1.0 else a


def gen_configs():
    arrays, traces = {}, {}

    def cfg1():
        x = md.Tensor([[0, 2, -2, 1], [-1, -1, -2, -2]], allow_grad=True)
        y = md.Tensor([[2, 3, 4, 5], [0, -1, -3, 2]], allow_grad=True)
        f = 2 * y * md.sin(x) - x ** 2
        f.backward(allow_higher_order=True)
        r = {"f": to_np(f).copy(), "dx": to_np(x.grad).copy(), "dy": to_np(y.grad).copy()}
        x.grad.backward()
        r.update(d2x=to_np(x.grad).copy(), d2xy=to_np(y.grad).copy())
        return r

    def cfg2(n=64):
        A, B = md.Tensor(f32(2, (n, n)), allow_grad=True), md.Tensor(f32(9, (n, n)), allow_grad=True)
        C = A @ B
        C.backward()
        return {"A": to_np(A), "B": to_np(B), "C": to_np(C).copy(), "dA": to_np(A.grad).copy(), "dB": to_np(B.grad).copy()}

    def cfg3(n=4096):
        x, y = md.Tensor(f32(3, (n,)), allow_grad=True), md.Tensor(f32(10, (n,)), allow_grad=True)
        loss = md.sum((md.sin(x) * y) ** 2)
        loss.backward()
        return {"x": to_np(x), "y": to_np(y), "loss": to_np(loss).copy(), "dx": to_np(x.grad).copy(), "dy": to_np(y.grad).copy()}

    def cfg4(b=64, di=32, do=48):
        rng = np.random.default_rng(4)
        X = md.Tensor(rng.standard_normal((b, di), dtype=np.float32))
        W = md.Tensor(rng.standard_normal((di, do), dtype=np.float32) / np.float32(8.0), allow_grad=True)
        bb = md.Tensor(rng.standard_normal((do,), dtype=np.float32), allow_grad=True)
        z = X @ W + bb
        loss = md.sum(md.where(z > 0, z, 0))
        loss.backward()
        return {"X": to_np(X), "W": to_np(W), "b": to_np(bb), "loss": to_np(loss).copy(), "dW": to_np(W.grad).copy(),
                "db": to_np(bb.grad).copy()}

    def cfg5(n=64):
        A, B = md.Tensor(f32(5, (n, n)), allow_grad=True), md.Tensor(f32(12, (n, n)), allow_grad=True)
        C = A @ B
        C.backward(allow_higher_order=True)
        first = (to_np(A.grad).copy(), to_np(B.grad).copy())
        A.grad.backward()
        return {"A": to_np(A), "B": to_np(B), "dA1": first[0], "dB1": first[1], "dA2": to_np(A.grad).copy(),
                "dB2": to_np(B.grad).copy()}

    for name, fn in (("cfg1", cfg1), ("cfg2", cfg2), ("cfg3", cfg3), ("cfg4", cfg4), ("cfg5", cfg5)):
        res, trace = traced(fn)
        for k, v in res.items():
            arrays[f"{name}/{k}"] = v
        traces[name] = trace
    return arrays, traces


def main():
    ops_arrays, ops_meta = gen_ops()
    cfg_arrays, traces = gen_configs()
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops_arrays)
    np.savez_compressed(os.path.join(HERE, "configs.npz"), **cfg_arrays)
    info = {"numpy": np.__version__, "python": sys.version.split()[0], "reference": "ahoynodnarb/minidiff @ /root/reference",
            "generator": "tests/golden/make_golden.py", "ops": ops_meta, "traces": traces}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(info, f, indent=0, sort_keys=True)
    n_raise = sum(1 for m in ops_meta.values() if "forward_raises" in m or "backward_raises" in m)
    print(f"{len(ops_meta)} op cases ({n_raise} recorded as raising in the reference), {len(traces)} config traces")


if __name__ == "__main__":
    main()
