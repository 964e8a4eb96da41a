"""Case table shared by the fixture generator (run against the REAL reference in
this container) and by the tests (run against this repo's tape over the NumPy
oracle table and over the HIP table). A case is engine-agnostic: it receives an
`md`-like namespace (reference minidiff, or minidiff_amd.tape engine) and host
input arrays, and returns the op's output Tensor.

Shapes follow the reference's own tests (tests/test_ops.py: randn(2,2,2,2),
matmul (10,30)@(30,20), dot on length-2 vectors) plus the broadcast / dtype
forms SURVEY.md §8c lists.
"""
import numpy as np


def _r(seed, shape, dtype=np.float64):
    return np.random.default_rng(seed).standard_normal(shape).astype(dtype)


def _case(name, inputs, fn, grads=None, dtypes=("float64", "float32")):
    return {"name": name, "inputs": inputs, "fn": fn, "grads": grads if grads is not None else [True] * len(inputs),
            "dtypes": dtypes}


S = (2, 2, 2, 2)
CASES = []
_add = CASES.append

for nm in ("absolute", "cos", "sin", "tan", "cosh", "sinh", "tanh", "exp", "copy", "ravel", "flatten", "atleast_1d",
           "atleast_2d", "atleast_3d"):
    _add(_case(nm, [_r(1, S)], lambda md, x, _n=nm: getattr(md, _n)(x)))
_add(_case("log", [np.abs(_r(2, S)) + 0.5], lambda md, x: md.log(x)))
_add(_case("sqrt", [np.abs(_r(3, S)) + 0.5], lambda md, x: md.sqrt(x)))
_add(_case("square", [_r(4, S)], lambda md, x: md.square(x)))
_add(_case("squeeze", [_r(5, (1, 2, 1, 2))], lambda md, x: md.squeeze(x)))
_add(_case("expand_dims", [_r(6, S)], lambda md, x: md.expand_dims(x, (1, 3))))
_add(_case("transpose", [_r(7, (2, 3, 4, 5))], lambda md, x: md.transpose(x)))
_add(_case("transpose_axes", [_r(8, (2, 3, 4, 5))],
           lambda md, x: md.transpose(x, axes=tuple(np.array([2, 0, 3, 1])))))
_add(_case("swapaxes", [_r(9, (2, 3, 4, 5))], lambda md, x: md.swapaxes(x, 1, 3)))
_add(_case("flip", [_r(10, S)], lambda md, x: md.flip(x, axis=(0, 2))))
_add(_case("flip_all", [_r(11, S)], lambda md, x: md.flip(x, axis=None)))
_add(_case("reshape", [_r(12, S)], lambda md, x: md.reshape(x, (4, 4))))
_add(_case("broadcast_to", [_r(13, S)], lambda md, x: md.broadcast_to(x, (4, 2, 2, 2, 2))))
_add(_case("clip", [_r(14, S)], lambda md, x: md.clip(x, -0.5, 0.7)))
_add(_case("clip_min_only", [_r(15, S)], lambda md, x: md.clip(x, 0.0, None)))
for ax in (None, (0,), (1, 3), (0, 1, 2, 3), ()):
    tag = "none" if ax is None else "_".join(map(str, ax)) or "empty"
    _add(_case(f"sum_{tag}", [_r(16, S)], lambda md, x, _a=ax: md.sum(x, axis=_a)))
    _add(_case(f"mean_{tag}", [_r(17, S)], lambda md, x, _a=ax: md.mean(x, axis=_a)))
    _add(_case(f"prod_{tag}", [_r(18, S)], lambda md, x, _a=ax: md.prod(x, axis=_a)))
    _add(_case(f"max_{tag}", [_r(19, S)], lambda md, x, _a=ax: md.max(x, axis=_a)))
    _add(_case(f"min_{tag}", [_r(20, S)], lambda md, x, _a=ax: md.min(x, axis=_a)))
    _add(_case(f"std_{tag}", [_r(21, S)], lambda md, x, _a=ax: md.std(x, axis=_a)))
_add(_case("sum_int_axis", [_r(22, S)], lambda md, x: md.sum(x, axis=1)))
_add(_case("max_int_axis", [_r(23, S)], lambda md, x: md.max(x, axis=2)))
_add(_case("min_int_axis", [_r(24, S)], lambda md, x: md.min(x, axis=2)))
_add(_case("mean_int_axis", [_r(25, S)], lambda md, x: md.mean(x, axis=0)))
_add(_case("sum_keepdims", [_r(26, S)], lambda md, x: md.sum(x, axis=(1, 2), keepdims=True)))

for nm in ("add", "subtract", "multiply", "true_divide"):
    _add(_case(nm, [_r(30, S), _r(31, S) + 3.0], lambda md, x, y, _n=nm: getattr(md, _n)(x, y)))
    _add(_case(nm + "_bcast", [_r(32, (2, 1, 3)), _r(33, (4, 1)) + 3.0], lambda md, x, y, _n=nm: getattr(md, _n)(x, y)))
    _add(_case(nm + "_scalar", [_r(34, S)], lambda md, x, _n=nm: getattr(md, _n)(x, 2.5)))
    _add(_case(nm + "_rscalar", [_r(35, S) + 3.0], lambda md, x, _n=nm: getattr(md, _n)(2, x)))
    _add(_case(nm + "_0d", [_r(36, S), np.array(1.75)], lambda md, x, y, _n=nm: getattr(md, _n)(x, y)))
_add(_case("power", [np.abs(_r(37, S)) + 0.5, _r(38, S)], lambda md, x, y: md.power(x, y)))
_add(_case("power_scalar2", [_r(39, S)], lambda md, x: x ** 2))
_add(_case("power_scalar_half", [np.abs(_r(40, S)) + 0.5], lambda md, x: x ** 0.5))
_add(_case("rpow", [_r(41, S)], lambda md, x: 2.0 ** x))
_add(_case("mod", [_r(42, S) * 3, np.abs(_r(43, S)) + 0.5], lambda md, x, y: md.mod(x, y)))
_add(_case("neg", [_r(44, S)], lambda md, x: -x))
_add(_case("dot", [_r(45, (2,)), _r(46, (2,))], lambda md, x, y: md.dot(x, y)))
_add(_case("matmul", [_r(47, (10, 30)), _r(48, (30, 20))], lambda md, x, y: md.matmul(x, y)))
_add(_case("matmul_chain", [_r(49, (6, 5)), _r(50, (5, 7)), _r(51, (7, 3))], lambda md, x, y, z: (x @ y) @ z))
_add(_case("tensordot", [_r(52, S), _r(53, S)], lambda md, x, y: md.tensordot(x, y)))
_add(_case("tensordot_axes", [_r(54, (2, 3, 4)), _r(55, (4, 3, 5))],
           lambda md, x, y: md.tensordot(x, y, axes=((1, 2), (1, 0)))))
_add(_case("where", [(_r(56, S) > 0), _r(57, S), _r(58, S)], lambda md, c, y, z: md.where(c, y, z),
           grads=[False, True, True]))
_add(_case("where_scalar_branch", [_r(59, S)], lambda md, x: md.where(x > 0, x, 0)))
_add(_case("getitem_intarray", [_r(60, S), np.array([1, 0, 1, 1])], lambda md, x, k: md.getitem(x, k), grads=[True, False]))
_add(_case("getitem_slice", [_r(61, S)], lambda md, x: x[:, 1, ::-1]))
_add(_case("getitem_int", [_r(62, S)], lambda md, x: x[1]))
_add(_case("astype", [_r(63, S)], lambda md, x: md.astype(x, md.float32) * 2.0))
_add(_case("shared_input", [_r(64, S)], lambda md, x: md.sin(x) * x + x))
_add(_case("chain_sin_mul_pow_sum", [_r(65, (64,)), _r(66, (64,))], lambda md, x, y: md.sum((md.sin(x) * y) ** 2)))
_add(_case("mlp", [_r(67, (16, 8)), _r(68, (8, 12)) / 4, _r(69, (12,))],
           lambda md, X, W, b: md.sum(md.where((X @ W + b) > 0, X @ W + b, 0)), grads=[False, True, True]))
# integer / bool forms (bit-exact)
_add(_case("int_arith", [np.array([[0, 2, -2, 1], [-1, -1, -2, -2]]), np.array([[2, 3, 4, 5], [0, -1, -3, 2]])],
           lambda md, x, y: 2 * y * md.sin(x) - x ** 2, dtypes=("int64",)))
_add(_case("int_floor_mod", [np.array([7, -7, 5, -5, 0, 9]), np.array([2, 2, -3, -3, 4, 0])],
           lambda md, x, y: md.floor_divide(x, y) * 10 + md.mod(x, y), grads=[False, False], dtypes=("int64",)))
_add(_case("compare_logic", [_r(70, S), _r(71, S)],
           lambda md, x, y: md.logical_xor(md.logical_and(x > y, x >= 0), md.logical_or(x < -1, md.not_equal(x, y))),
           grads=[False, False]))
_add(_case("argmax_keepdims", [_r(72, (3, 4, 5))], lambda md, x: md.argmax(x, axis=1, keepdims=True), grads=[False]))
_add(_case("any_all", [_r(73, (3, 4)) > 0], lambda md, x: md.logical_and(md.any(x), md.all(x)), grads=[False],
           dtypes=("bool",)))

# storage-only dtypes of the reference table (backend/numpy.py:188-200): float16, int8/16, uint8/16/32 — integers bit-exact
# (NumPy's wrap-around included), float16 within its own precision
_NI = ("int8", "int16", "uint8", "uint16", "uint32")
_ia = np.array([[100, -100, 27, -128, 127, 5], [3, -7, 90, 64, -64, 1]])
_ib = np.array([[3, 2, 9, 1, 1, -4], [50, 13, 2, 2, -3, 120]])
_add(_case("narrow_int_wrap", [_ia, _ib], lambda md, x, y: x * y + x - y, grads=[False, False], dtypes=_NI))
_add(_case("narrow_int_divmod", [np.array([7, 100, 5, 120, 0, 9]), np.array([2, 7, 3, 11, 4, 5])],
           lambda md, x, y: md.floor_divide(x, y) * 10 + md.mod(x, y) + md.power(y, 2), grads=[False, False], dtypes=_NI))
_add(_case("narrow_int_where_max", [_ia, _ib], lambda md, x, y: md.where(x > y, x, y) - md.max(x, axis=(1,), keepdims=True),
           grads=[False, False], dtypes=_NI))
_add(_case("narrow_int_sum", [_ia], lambda md, x: md.sum(x, axis=(0,)), grads=[False], dtypes=_NI))
_add(_case("narrow_int_true_divide", [_ia, _ib], lambda md, x, y: md.true_divide(x, y), grads=[False, False], dtypes=("int8", "uint8", "int16")))
_add(_case("narrow_int_mixed", [_ia.astype(np.int8), _ib.astype(np.int64)], lambda md, x, y: x + y, grads=[False, False], dtypes=("asis",)))
_add(_case("narrow_int_getitem", [np.arange(24).reshape(4, 6), np.array([3, 0, 3, 1])], lambda md, x, i: x[i], grads=[False, False],
           dtypes=("int16", "uint8")))
_add(_case("narrow_f16_chain", [_r(80, (3, 5)), _r(81, (3, 5))], lambda md, x, y: md.sin(x) * y + x ** 2, dtypes=("float16",)))
_add(_case("narrow_f16_broadcast_sum", [_r(82, (4, 6)), _r(83, (6,))], lambda md, x, b: md.sum(md.where(x + b > 0, x + b, 0), axis=(0,)),
           dtypes=("float16",)))
_add(_case("narrow_f16_matmul", [_r(84, (4, 8)), _r(85, (8, 3)) / 4], lambda md, X, W: X @ W, dtypes=("float16",)))
_add(_case("narrow_f16_getitem", [_r(86, (5, 4)), np.array([4, 1, 1, 0])], lambda md, x, i: x[i], grads=[True, False], dtypes=("float16",)))


def loss_of(md, out):
    """tests/test_ops.py:42-45 — half squared distance to zero."""
    expected = md.zeros_like(out)
    return md.sum((expected - out) ** 2) / 2
