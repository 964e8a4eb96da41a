"""Run-time specialisation of fused programs: the generated HIP source must
compile for gfx950 (hiprtc, no device needed) for every operator and epilogue the
interpreter accepts. Numerical parity of the compiled kernels is covered on the
GPU by running the lazy suites with MDHIP_JIT_MIN=1 (tests/test_lazy_fusion.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from minidiff_amd import _capi


@pytest.fixture(scope="module")
def product():
    if not os.path.exists(_capi.PRODUCT_LIB):
        pytest.skip("libmdhip.so not built")
    lib = _capi.Library(_capi.PRODUCT_LIB)  # dlopen only; compile-only probe needs no GPU
    log = C.create_string_buffer(4096)
    return lib, log


def _programs(nd):
    from minidiff_amd import lazy as lz
    rng = np.random.default_rng(0)
    x = nd.asarray(rng.standard_normal((8, 16)).astype(np.float32))
    y = nd.asarray(rng.standard_normal((8, 16)).astype(np.float32))
    b = nd.asarray(rng.standard_normal((16,)).astype(np.float32))
    m = nd.asarray(rng.integers(0, 2, (8, 16)).astype(bool))
    xd = nd.asarray(rng.standard_normal((8, 16)))
    xi = nd.asarray(rng.integers(-3, 3, (8, 16)))
    progs = []
    un = ["absolute", "sign", "ceil", "floor", "sin", "cos", "tan", "sinh", "cosh", "tanh", "exp", "log", "sqrt",
          "logical_not", "negative", "isnan"]
    bi = ["add", "subtract", "multiply", "true_divide", "power", "mod", "floor_divide", "maximum", "minimum", "less",
          "less_equal", "greater", "greater_equal", "equal", "not_equal", "logical_and", "logical_or", "logical_xor"]
    for name in un:
        progs.append((name, getattr(nd, name)(nd.multiply(x, 1.5))))
    for name in bi:
        progs.append((name, getattr(nd, name)(nd.add(x, b), y)))
        progs.append((name + "/const-left", getattr(nd, name)(2.0, nd.sin(x))))
    progs.append(("where", nd.where(nd.greater(x, 0), nd.multiply(x, m), 0.25)))
    progs.append(("f64+int leaf", nd.add(nd.exp(xd), xi)))
    # every leaf read mode, with the hoisted sin / cos on each of them: unit stride (x), row-invariant (b), column
    # broadcast (col), one device element behind a stride-0 view (seed)
    col = nd.asarray(rng.standard_normal((8, 1)).astype(np.float32))
    seed = nd.broadcast_to(nd.asarray(np.float32(0.5)), (8, 16))
    progs.append(("leaf modes", nd.add(nd.add(nd.multiply(nd.sin(x), nd.cos(x)), nd.multiply(nd.sin(b), nd.cos(col))),
                                       nd.multiply(nd.multiply(nd.sin(seed), nd.cos(seed)), nd.sin(col)))))
    progs.append(("cfg3 x.grad", nd.multiply(nd.multiply(nd.multiply(nd.multiply(
        nd.broadcast_to(nd.asarray(np.float32(1)), (8, 16)), 2), nd.power(nd.multiply(nd.sin(x), y), 1)), y), nd.cos(x))))
    out = []
    for name, arr in progs:
        assert arr._expr is not None, name
        prog, keep = lz.build_program(arr._expr, arr.shape)
        out.append((name, prog, keep, arr.dtype == np.bool_))
    return out


def test_generated_kernels_compile(lib, on_gpu, product):
    from minidiff_amd import ndarray as nd
    plib, log = product
    prev = nd.set_lazy(True)
    try:
        progs = _programs(nd)
    finally:
        nd.set_lazy(prev)
    n = 0
    for name, prog, keep, is_bool in progs:
        plib.vm_jit_probe(prog, 0, 0, int(is_bool), log, len(log))
        n += 1
        if not is_bool and name in ("sin", "multiply", "where", "f64+int leaf", "cfg3 x.grad", "leaf modes"):   # epilogues: a subset keeps this quick
            for rop in (_capi.R_SUM, _capi.R_PROD, _capi.R_MAX, _capi.R_MIN):
                plib.vm_jit_probe(prog, 1, rop, 0, log, len(log))
                plib.vm_jit_probe(prog, 2, rop, 0, log, len(log))
                n += 2
            # sweep-style column reduce, without / with the evaluated value stored (mdhip_vm_eval_reduce_cols)
            plib.vm_jit_probe(prog, 3, _capi.R_SUM, 0, log, len(log))
            plib.vm_jit_probe(prog, 4, _capi.R_SUM, 0, log, len(log))
            plib.vm_jit_probe(prog, 4, _capi.R_MAX, 0, log, len(log))
            n += 3
    assert n > 110


def test_generated_multi_output_kernel_compiles(lib, on_gpu, product):
    """The multi-output form (mdhip_vm_eval_multi): merged leaf table, one inlined body per
    program, shared immediates array."""
    from minidiff_amd import lazy as lz, ndarray as nd
    plib, log = product
    prev = nd.set_lazy(True)
    try:
        rng = np.random.default_rng(1)
        x = nd.asarray(rng.standard_normal((8, 16)).astype(np.float32))
        y = nd.asarray(rng.standard_normal((8, 16)).astype(np.float32))
        b = nd.asarray(rng.standard_normal((16,)).astype(np.float32))
        s = nd.multiply(nd.sin(x), y)
        arrs = [nd.multiply(nd.multiply(s, 2.0), nd.cos(x)), nd.add(nd.multiply(s, nd.sin(x)), b),
                nd.where(nd.greater(y, 0.25), nd.exp(x), -1.5), nd.subtract(2.0, x)]
        for n in (2, 3, 4):
            progs = (_capi.VmProgram * n)()
            keep = []
            for k in range(n):
                prog, kp = lz.build_program(arrs[k]._expr, arrs[k].shape)
                progs[k] = prog
                keep.append(kp)
            plib.vm_jit_probe_multi(progs, n, log, len(log))
        one = (_capi.VmProgram * 1)()
        one[0] = lz.build_program(arrs[0]._expr, arrs[0].shape)[0]
        with pytest.raises(ValueError):
            plib.vm_jit_probe_multi(one, 1, log, len(log))
    finally:
        nd.set_lazy(prev)
