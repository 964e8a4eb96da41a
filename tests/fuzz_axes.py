"""Focused differential fuzz of the three- / four-axis vector kernels (elementwise.hip k_ew_axes, fusion.hip k_vm_eval_axes) against
NumPy: random 3-D / 4-D iteration spaces around the kernels' conditions (inner extent a multiple of 4 or not, >= 65536 elements or
not), every operand independently broadcast along random axes, sliced (aligned and misaligned offsets), axis-swapped or a scalar;
unary / binary / comparison / where calls, eager and fused.   python tests/fuzz_axes.py [n_cases] [seed]
Runs on whatever library tests/conftest.py binds (the HIP library on a GPU box, the CPU double elsewhere)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conftest  # noqa: E402

lib, ON_GPU = conftest.bound_library()
from minidiff_amd import ndarray as nd  # noqa: E402

DTYPES = (np.float32, np.float64, np.int32, np.int64)
BINARY = ("add", "subtract", "multiply", "maximum", "minimum")
COMPARE = ("greater", "less_equal", "equal", "not_equal")
UNARY = ("negative", "absolute", "sign")


def space(rng):
    nd_ = int(rng.choice([3, 3, 4]))
    inner = int(rng.choice([4, 8, 64, 128, 132, 256, 130, 63]))
    outer = [int(rng.choice([1, 2, 3, 5, 16, 33, 64])) for _ in range(nd_ - 1)]
    while np.prod(outer) * inner > (1 << 21):
        outer[int(np.argmax(outer))] = max(1, outer[int(np.argmax(outer))] // 2)
    return tuple(outer) + (inner,)


def operand(rng, shape, dt, allow_scalar=True):
    """(host array, device array) broadcastable to `shape`: random broadcast axes, then maybe a slice of a larger base / an axis swap."""
    r = rng.random()
    if allow_scalar and r < 0.1:
        v = float(rng.integers(-5, 6)) if np.dtype(dt).kind == "f" else int(rng.integers(-5, 6))
        return v, v
    shp = [1 if rng.random() < 0.35 else s for s in shape]
    if rng.random() < 0.15:
        shp = shp[1:]                                      # fewer dimensions
    mk = (lambda s: (rng.standard_normal(s) * 8).astype(dt)) if np.dtype(dt).kind == "f" else (lambda s: rng.integers(-9, 10, s).astype(dt))
    if np.dtype(dt) == np.bool_:
        mk = lambda s: rng.random(s) < 0.5
    kind = rng.random()
    if kind < 0.45:
        h = mk(shp)
        return h, nd.asarray(h)
    if kind < 0.85:                                        # slice of a larger base: offsets 0 / 1 / 2 / 4 along each axis
        pads = [int(rng.choice([0, 1, 2, 4])) for _ in shp]
        base = mk([s + 2 * p for s, p in zip(shp, pads)])
        sl = tuple(slice(p, p + s) for s, p in zip(shp, pads))
        return base[sl], nd.asarray(base)[sl]
    if len(shp) >= 2:                                      # two outer axes swapped (the inner one stays contiguous)
        i, j = sorted(rng.choice(len(shp) - 1, 2, replace=False)) if len(shp) >= 3 else (0, 0)
        sw = list(shp)
        sw[i], sw[j] = sw[j], sw[i]
        base = mk(sw)
        perm = list(range(len(shp)))
        perm[i], perm[j] = perm[j], perm[i]
        return base.transpose(perm), nd.transpose(nd.asarray(base), perm)
    h = mk(shp)
    return h, nd.asarray(h)


def same(got, want, what):
    got = got.get() if hasattr(got, "get") else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype, (what, got.shape, want.shape, got.dtype, want.dtype)
    assert np.array_equal(got, want, equal_nan=True), what


def one(rng, lazy):
    shape = space(rng)
    dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
    full_h, full_d = operand(rng, shape, dt, allow_scalar=False)
    if np.shape(full_h) != shape and rng.random() < 0.8:   # make sure one operand spans the space most of the time
        full_h = (rng.standard_normal(shape) * 8).astype(dt) if np.dtype(dt).kind == "f" else rng.integers(-9, 10, shape).astype(dt)
        full_d = nd.asarray(full_h)
    bh, bd = operand(rng, shape, dt)
    k = rng.random()
    fin = nd.materialize if lazy else (lambda x: x)
    with np.errstate(all="ignore"):
        if k < 0.2:
            name = UNARY[int(rng.integers(0, len(UNARY)))]
            same(fin(getattr(nd, name)(full_d)), getattr(np, name)(full_h), (name, shape, dt))
        elif k < 0.55:
            name = BINARY[int(rng.integers(0, len(BINARY)))]
            if rng.random() < 0.5:
                same(fin(getattr(nd, name)(full_d, bd)), getattr(np, name)(full_h, bh), (name, shape, np.shape(bh), dt))
            else:
                same(fin(getattr(nd, name)(bd, full_d)), getattr(np, name)(bh, full_h), (name, np.shape(bh), shape, dt))
        elif k < 0.75:
            name = COMPARE[int(rng.integers(0, len(COMPARE)))]
            same(fin(getattr(nd, name)(full_d, bd)), getattr(np, name)(full_h, bh), (name, shape, np.shape(bh), dt))
        elif k < 0.9:
            mh, md = operand(rng, shape, np.bool_, allow_scalar=False)
            same(fin(nd.where(md, full_d, bd)), np.where(mh, full_h, bh), ("where", shape, np.shape(mh), np.shape(bh), dt))
        else:                                              # a chain: in lazy mode one fused program
            ch, cd = operand(rng, shape, dt)
            same(fin(nd.subtract(nd.multiply(full_d, bd), cd)), np.subtract(np.multiply(full_h, bh), ch), ("chain", shape, np.shape(bh), np.shape(ch), dt))


def main(n=2000, seed=0):
    rng = np.random.default_rng(seed)
    fails = 0
    for lazy in (False, True):
        prev = nd.set_lazy(lazy)
        try:
            for i in range(n // 2):
                try:
                    one(rng, lazy)
                except AssertionError as e:
                    fails += 1
                    print(f"FAIL lazy={lazy} case {i}: {e}", flush=True)
                if (i + 1) % 250 == 0:
                    print(f"  .. lazy={lazy} {i + 1} cases, {fails} failures", flush=True)
        finally:
            nd.set_lazy(prev)
    print(f"fuzz_axes: {n} cases, seed {seed}, gpu={ON_GPU}: {fails} failures", flush=True)
    return fails


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sys.exit(1 if main(int(args[0]) if args else 2000, int(args[1]) if len(args) > 1 else 0) else 0)
