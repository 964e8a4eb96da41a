"""Hostile descriptors through the C-ABI (include/mdhip.h), on the CPU double — and, when the sanitizer build exists, on the ASan +
UBSan double in a subprocess: out-of-range ndim / dtype / op codes, null data, negative extents, mismatched shapes, bad reduction
masks, index plans with bad counts and dtypes, non-3-D matmul operands. The validation code (csrc/md_common.h, md_dispatch.h,
md_narrow.h) is shared with the product library, so "an error code, never a crash" is checked for both. A C caller of the
boundary gets MDHIP_E*; nothing here goes through the Python shim's own argument checks."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _hostile(lib_path, device=False):
    from minidiff_amd import _capi
    raw = C.CDLL(lib_path)
    A = _capi.ArrayDesc
    P = _capi.IndexPlan
    raw.mdhip_init(0)
    if device:      # the product library: the same descriptors over DEVICE memory
        from minidiff_amd import ndarray as nd
        dbuf = nd.zeros((4096,), np.float64)
        didx = nd.zeros((16,), np.int64)
        ptr, idx_addr = dbuf.ptr, didx.ptr
    else:
        buf = np.zeros(4096, np.float64)
        idx = np.zeros(16, np.int64)
        ptr, idx_addr = buf.ctypes.data, idx.ctypes.data

    def set_idx0(v):
        if device:
            didx[0] = v
            nd._lib().sync()
        else:
            idx[0] = v

    def arr(shape=(4, 4), dtype=_capi.F32, strides=None, data=ptr, ndim=None, scalar=0):
        d = A()
        d.data = data
        d.dtype = dtype
        d.ndim = len(shape) if ndim is None else ndim
        for i, n in enumerate(shape[:8]):
            d.shape[i] = n
        st = strides
        if st is None:
            st, acc = [], 1
            for n in reversed(shape):
                st.insert(0, acc)
                acc *= max(n, 1)
        for i, v in enumerate(st[:8]):
            d.strides[i] = v
        d.is_scalar = scalar
        return d

    n_err = 0

    def expect_error(name, rc):
        nonlocal n_err
        assert rc != 0, f"{name}: accepted"
        assert rc in (1, 2, 3, 4, 5), (name, rc)
        assert raw.mdhip_last_error is not None
        n_err += 1

    good = arr()
    ref = C.byref
    for bad_nd in (-1, 9, 100):
        expect_error(f"unary ndim {bad_nd}", raw.mdhip_unary(0, ref(arr(ndim=bad_nd)), ref(good)))
        expect_error(f"binary out ndim {bad_nd}", raw.mdhip_binary(0, ref(good), ref(good), ref(arr(ndim=bad_nd)), _capi.F32))
        expect_error(f"reduce ndim {bad_nd}", raw.mdhip_reduce(0, ref(arr(ndim=bad_nd)), ref(arr((1, 1))), 3))
    for bad_dt in (-1, 12, 99):
        expect_error(f"unary dtype {bad_dt}", raw.mdhip_unary(0, ref(arr(dtype=bad_dt)), ref(good)))
        expect_error(f"binary cdt {bad_dt}", raw.mdhip_binary(0, ref(good), ref(good), ref(good), bad_dt))
        expect_error(f"convert dtype {bad_dt}", raw.mdhip_convert(ref(good), ref(arr(dtype=bad_dt))))
        expect_error(f"fill dtype {bad_dt}", raw.mdhip_fill(ref(arr(dtype=bad_dt)), ref(arr((), scalar=1))))
    for bad_op in (-1, 18, 1000):
        expect_error(f"unary op {bad_op}", raw.mdhip_unary(bad_op, ref(good), ref(good)))
        expect_error(f"binary op {bad_op}", raw.mdhip_binary(bad_op, ref(good), ref(good), ref(good), _capi.F32))
        expect_error(f"reduce op {bad_op}", raw.mdhip_reduce(bad_op, ref(good), ref(arr((1, 1))), 3))
    expect_error("null data", raw.mdhip_unary(0, ref(arr(data=None)), ref(good)))
    expect_error("null out", raw.mdhip_unary(0, ref(good), ref(arr(data=None))))
    expect_error("negative extent", raw.mdhip_unary(0, ref(arr((4, -4))), ref(good)))
    expect_error("shape mismatch", raw.mdhip_binary(0, ref(arr((4, 4))), ref(arr((3, 4))), ref(good), _capi.F32))
    expect_error("out shape mismatch", raw.mdhip_binary(0, ref(good), ref(good), ref(arr((4, 5))), _capi.F32))
    expect_error("where shape", raw.mdhip_where(ref(arr((4, 4), dtype=_capi.BOOL)), ref(arr((2, 4))), ref(good), ref(good)))
    expect_error("reduce mask beyond ndim", raw.mdhip_reduce(0, ref(good), ref(arr((1, 1))), 1 << 5))
    expect_error("reduce out shape", raw.mdhip_reduce(0, ref(good), ref(arr((4, 4))), 3))
    expect_error("matmul 2-d", raw.mdhip_matmul(ref(good), ref(good), ref(good)))
    expect_error("matmul inner", raw.mdhip_matmul(ref(arr((1, 4, 3))), ref(arr((1, 4, 4))), ref(arr((1, 4, 4)))))
    expect_error("matmul dtypes", raw.mdhip_matmul(ref(arr((1, 4, 4))), ref(arr((1, 4, 4), dtype=_capi.F64)), ref(arr((1, 4, 4)))))
    expect_error("matmul bool", raw.mdhip_matmul(ref(arr((1, 4, 4), dtype=_capi.BOOL)), ref(arr((1, 4, 4), dtype=_capi.BOOL)), ref(arr((1, 4, 4), dtype=_capi.BOOL))))
    # index plans
    def plan(ndim=1, n_idx=1, shape=(4,), idx_dtype=_capi.I64, extent=4, idx_ptr=idx_addr):
        p = P()
        p.ndim, p.n_idx = ndim, n_idx
        for i, n in enumerate(shape[:8]):
            p.shape[i] = n
        for k in range(min(max(n_idx, 0), 8)):
            p.idx_ptr[k] = idx_ptr
            p.idx_dtype[k] = idx_dtype
            p.idx_extent[k] = extent
            p.idx_mult[k] = 1
            p.idx_strides[k][0] = 1
        return p

    out1 = arr((4,))
    for name, pl in (("plan ndim 9", plan(ndim=9)), ("plan ndim -1", plan(ndim=-1)), ("plan n_idx 9", plan(n_idx=9)), ("plan n_idx -1", plan(n_idx=-1)),
                     ("plan idx dtype f32", plan(idx_dtype=_capi.F32)), ("plan idx dtype 77", plan(idx_dtype=77)), ("plan null idx", plan(idx_ptr=None)),
                     ("plan negative extent", plan(shape=(-4,))), ("plan zero index extent", plan(extent=0))):
        expect_error("gather " + name, raw.mdhip_gather(ref(pl), C.c_void_p(ptr), _capi.F32, ref(out1)))
        expect_error("scatter " + name, raw.mdhip_scatter(ref(pl), C.c_void_p(ptr), _capi.F32, ref(out1), 0))
    expect_error("gather out ndim", raw.mdhip_gather(ref(plan()), C.c_void_p(ptr), _capi.F32, ref(good)))
    expect_error("gather dtype", raw.mdhip_gather(ref(plan()), C.c_void_p(ptr), 55, ref(out1)))
    expect_error("scatter mode", raw.mdhip_scatter(ref(plan()), C.c_void_p(ptr), _capi.F32, ref(out1), 7))
    expect_error("scatter value dtype", raw.mdhip_scatter(ref(plan()), C.c_void_p(ptr), _capi.F32, ref(arr((4,), dtype=_capi.F64)), 0))
    expect_error("scatter null value", raw.mdhip_scatter(ref(plan()), C.c_void_p(ptr), _capi.F32, None, 0))
    set_idx0(9)
    expect_error("gather index out of range", raw.mdhip_gather(ref(plan()), C.c_void_p(ptr), _capi.F32, ref(out1)))
    expect_error("scatter index out of range", raw.mdhip_scatter(ref(plan()), C.c_void_p(ptr), _capi.F32, ref(out1), 1))
    set_idx0(0)
    expect_error("unknown option", raw.mdhip_debug_set_option(b"no_such_option", C.c_int64(1)))
    expect_error("null option", raw.mdhip_debug_set_option(None, C.c_int64(1)))
    # and the library still works afterwards
    x = np.arange(16, dtype=np.float32)
    if device:
        dxa, dya = nd.asarray(x), nd.zeros((16,), np.float32)
        dx, dy = arr((4, 4), data=dxa.ptr), arr((4, 4), data=dya.ptr)
        assert raw.mdhip_unary(_capi.U_NEG, ref(dx), ref(dy)) == 0 and np.array_equal(dya.get(), -x)
    else:
        y = np.zeros(16, np.float32)
        dx, dy = arr((4, 4), data=x.ctypes.data), arr((4, 4), data=y.ctypes.data)
        assert raw.mdhip_unary(_capi.U_NEG, ref(dx), ref(dy)) == 0 and np.array_equal(y, -x)
    return n_err


def _random_vm_programs(lib_path, n_programs, seed, device=False):
    """Random bytecode for the fused-expression interpreter (include/mdhip.h mdhip_vm_program, csrc/md_vm.h): arbitrary ctrl words,
    instruction / leaf counts and compute dtypes. md_vm_check must reject whatever could underflow or overrun the four-slot stack or
    name a missing leaf; what it accepts must run to completion on valid leaves. Returns (accepted, rejected)."""
    from minidiff_amd import _capi
    raw = C.CDLL(lib_path)
    raw.mdhip_init(0)
    rng = np.random.default_rng(seed)
    leaves = [np.ascontiguousarray(rng.standard_normal((6, 5)).astype(np.float32)) for _ in range(_capi.VM_MAX_LEAVES)]
    out = np.zeros((6, 5), np.float32)
    if device:
        from minidiff_amd import ndarray as nd
        keep = [nd.asarray(x) for x in leaves] + [nd.asarray(out)]
        leaf_ptr, out_ptr = [k.ptr for k in keep[:-1]], keep[-1].ptr
    else:
        leaf_ptr, out_ptr = [x.ctypes.data for x in leaves], out.ctypes.data
    outd = _capi.ArrayDesc()
    outd.data, outd.dtype, outd.ndim = out_ptr, _capi.F32, 2
    outd.shape[0], outd.shape[1], outd.strides[0], outd.strides[1] = 6, 5, 5, 1
    ok = bad = 0
    for _ in range(n_programs):
        pr = _capi.VmProgram()
        style = rng.integers(0, 3)
        pr.n_instr = int(rng.integers(-2, _capi.VM_MAX_INSTR + 3)) if style == 0 else int(rng.integers(1, 12))
        pr.n_leaves = int(rng.integers(-1, _capi.VM_MAX_LEAVES + 2)) if style == 0 else int(rng.integers(1, _capi.VM_MAX_LEAVES + 1))
        pr.compute_dtype = int(rng.choice([_capi.F32, _capi.F32, _capi.F32, _capi.F64, _capi.I32, 77]))
        for i in range(_capi.VM_MAX_INSTR):
            pr.ctrl[i] = int(rng.integers(0, 1 << 32))
            pr.imm[i] = float(rng.standard_normal())
        if style == 2:      # a stack-consistent program (so that the interpreter itself runs), one word in ten corrupted afterwards
            words, depth, nl = [], 0, max(pr.n_leaves, 1)
            ctrl = lambda kind, op, ls, ll, rs, rl: kind | (op << 3) | (ls << 8) | (ll << 10) | (rs << 13) | (rl << 15)   # noqa: E731
            UN = [u for u in range(18) if u != _capi.U_INVERT]
            for _i in range(int(rng.integers(1, 30))):
                choices = ["push"] if depth == 0 else (["push"] if depth < 4 else []) + ["unary", "binary_leaf"] + (["binary_stack"] if depth >= 2 else []) + (["where"] if depth >= 3 else [])
                c = str(rng.choice(choices))
                if c == "push":
                    words.append(ctrl(0, 0, 0, 0, int(rng.integers(1, 3)), int(rng.integers(0, nl)))); depth += 1
                elif c == "unary":
                    words.append(ctrl(1, int(rng.choice(UN)), 0, 0, 0, 0))
                elif c == "binary_leaf":
                    words.append(ctrl(2, int(rng.integers(0, 18)), 0, 0, int(rng.integers(1, 3)), int(rng.integers(0, nl))))
                elif c == "binary_stack":
                    words.append(ctrl(2, int(rng.integers(0, 18)), 0, 0, 0, 0)); depth -= 1
                else:
                    words.append(ctrl(3, 0, 0, 0, 0, 0)); depth -= 2
            while depth > 1:
                words.append(ctrl(2, int(rng.integers(0, 9)), 0, 0, 0, 0)); depth -= 1
            words = words[:_capi.VM_MAX_INSTR]
            pr.n_instr = len(words)
            for i, w in enumerate(words):
                pr.ctrl[i] = w if rng.random() > 0.1 else w ^ (1 << int(rng.integers(0, 20)))
        for k in range(_capi.VM_MAX_LEAVES):
            d = pr.leaves[k]
            d.data, d.dtype, d.ndim = leaf_ptr[k], _capi.F32, 2
            d.shape[0], d.shape[1], d.strides[0], d.strides[1] = 6, 5, 5, 1
        if pr.compute_dtype == _capi.F64:
            continue        # (float32 leaves and out: keep the accepted programs type-consistent)
        rc = raw.mdhip_vm_eval(C.byref(pr), C.byref(outd))
        assert rc in (0, 1, 2, 3, 4, 5), rc
        ok += rc == 0
        bad += rc != 0
    if device:
        nd._lib().sync()
    return ok, bad


def test_random_vm_programs_are_rejected_or_run(lib, on_gpu):
    if on_gpu:
        pytest.skip("host-memory descriptors: CPU double only")
    ok, bad = _random_vm_programs(os.path.join(ROOT, "oracle", "_build", "libmdhip_host.so"), 4000, 1)
    assert ok > 20 and bad > 1000, (ok, bad)


def test_hostile_descriptors_get_error_codes(lib, on_gpu):
    if on_gpu:
        pytest.skip("host-memory descriptors: CPU double only")
    n = _hostile(os.path.join(ROOT, "oracle", "_build", "libmdhip_host.so"))
    assert n >= 60


@pytest.mark.gpu
def test_hostile_descriptors_get_error_codes_gpu(lib, on_gpu):
    """The same calls against the PRODUCT library with device memory behind the descriptors: nothing may reach a kernel."""
    assert on_gpu
    n = _hostile(os.path.join(ROOT, "minidiff_amd", "libmdhip.so"), device=True)
    assert n >= 60


@pytest.mark.gpu
def test_random_vm_programs_are_rejected_or_run_gpu(lib, on_gpu):
    assert on_gpu
    ok, bad = _random_vm_programs(os.path.join(ROOT, "minidiff_amd", "libmdhip.so"), 3000, 3, device=True)
    assert ok > 100 and bad > 1000, (ok, bad)


def test_hostile_descriptors_on_the_sanitized_double(on_gpu):
    if on_gpu:
        pytest.skip("CPU-double check")
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asan_so = os.path.join(ROOT, "oracle", "_build", "libmdhip_host_asan.so")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {HERE!r}); import test_capi_negative as t; "
            f"print('errors', t._hostile({asan_so!r})); print('vm', t._random_vm_programs({asan_so!r}, 3000, 2))")
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "errors" in p.stdout and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stdout[-1500:] + p.stderr[-3000:]
