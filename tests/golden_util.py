import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
if GOLD not in sys.path:
    sys.path.insert(0, GOLD)
import cases  # noqa: E402

_CACHE = {}


def golden():
    if not _CACHE:
        with open(os.path.join(GOLD, "golden.json")) as f:
            _CACHE["info"] = json.load(f)
        _CACHE["ops"] = dict(np.load(os.path.join(GOLD, "ops.npz")))
        _CACHE["cfg"] = dict(np.load(os.path.join(GOLD, "configs.npz")))
    return _CACHE


def case_keys():
    return [f"{c['name']}@{dt}" for c in cases.CASES for dt in c["dtypes"]]


def case_by_name(name):
    for c in cases.CASES:
        if c["name"] == name:
            return c
    raise KeyError(name)


def to_np(t):
    d = t._data
    return d.get() if hasattr(d, "get") else np.asarray(d)


def rel_err(got, exp):
    got = np.asarray(got, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    mask = np.isfinite(exp)
    if not mask.any():
        return 0.0
    assert np.array_equal(np.isnan(got), np.isnan(exp)), "NaN pattern differs"
    assert np.array_equal(got[np.isinf(exp)], exp[np.isinf(exp)]), "inf pattern differs"
    scale = np.abs(exp[mask]).max()
    return float(np.abs(got[mask] - exp[mask]).max() / (scale if scale > 0 else 1.0))


def run_case_on(md, key, exact):
    """Run one golden case on engine `md`; compare with the reference's record.
    exact=True: bit-for-bit (NumPy oracle). exact=False: dtype/shape exact, ints
    exact, floats within 1e-5 (f32) / 1e-12 (f64) norm-wise."""
    g = golden()
    meta, arrs = g["info"]["ops"][key], g["ops"]
    name, dtype = key.split("@")
    case = case_by_name(name)
    ins = [arrs[f"{key}/in{i}"] for i in range(meta["n_inputs"])]

    def compare(tag, got, exp):
        assert got.dtype == exp.dtype, (key, tag, got.dtype, exp.dtype)
        assert got.shape == exp.shape, (key, tag, got.shape, exp.shape)
        if exact or exp.dtype.kind in "biu":
            assert np.array_equal(got, exp, equal_nan=exp.dtype.kind == "f"), (key, tag)
        else:
            # bound follows the precision the case COMPUTES in (f32 inputs may yield f64 grads)
            tol = 4e-3 if (exp.dtype == np.float16 or dtype == "float16") else 1e-5 if (exp.dtype == np.float32 or dtype == "float32") else 1e-12
            e = rel_err(got, exp)
            assert e <= tol, (key, tag, e)

    def build():
        tensors = [md.Tensor(a.copy(), allow_grad=gflag) for a, gflag in zip(ins, case["grads"])]
        return tensors, case["fn"](md, *tensors)

    if "forward_raises" in meta:
        try:
            build()
        except Exception as e:
            assert type(e).__name__ == meta["forward_raises"], (key, type(e).__name__, meta["forward_raises"])
            return
        raise AssertionError(f"{key}: the reference raises {meta['forward_raises']} in forward, this engine did not")
    tensors, out = build()
    compare("forward", to_np(out), arrs[f"{key}/forward"])
    if "backward_raises" in meta:
        try:
            cases.loss_of(md, out).backward()
        except Exception as e:
            assert type(e).__name__ == meta["backward_raises"], (key, type(e).__name__, meta["backward_raises"])
            return
        raise AssertionError(f"{key}: the reference raises {meta['backward_raises']} in backward, this engine did not")
    if "grad_present" not in meta:
        return
    loss = cases.loss_of(md, out)
    compare("loss", to_np(loss), arrs[f"{key}/loss"])
    loss.backward()
    for i, present in enumerate(meta["grad_present"]):
        if not present:
            assert tensors[i].grad is None, (key, i)
            continue
        assert tensors[i].grad is not None, (key, i)
        compare(f"grad{i}", to_np(tensors[i].grad), arrs[f"{key}/grad{i}"])
