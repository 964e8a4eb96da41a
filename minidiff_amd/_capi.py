"""ctypes binding of the libmdhip C-ABI (include/mdhip.h).

This is the whole Python<->native seam: struct layouts, prototypes and the
status-code -> exception mapping. The product library is
``minidiff_amd/libmdhip.so`` (HIP, gfx950); it is the only thing
:func:`load` will pick up by itself, and a missing or non-HIP library is a
hard ImportError — there is no CPU fallback in the product path. Tests may
hand an explicit path (the CPU test double under ``oracle/``) to
:func:`use_library`.

Reference boundary this replaces: the NumPy alias table
``minidiff/backend/numpy.py:14-206`` (every attribute there ends in one of the
entry points declared below).
"""
from __future__ import annotations

import ctypes as C
import os

MAX_NDIM = 8
UID_BYTES = 128

# dtype codes (mdhip.h)
BOOL, I32, I64, F32, F64 = range(5)
I8, I16, U8, U16, U32, U64, F16 = range(5, 12)   # storage-only dtypes (include/mdhip.h): moved and converted, never computed in

# op codes — keep in the order of the enums in mdhip.h
U_COPY, U_ABS, U_NEG, U_SIGN, U_CEIL, U_FLOOR, U_SIN, U_COS, U_TAN, U_SINH, U_COSH, U_TANH, \
    U_EXP, U_LOG, U_SQRT, U_LOGICAL_NOT, U_INVERT, U_ISNAN = range(18)
B_ADD, B_SUB, B_MUL, B_TRUE_DIV, B_FLOOR_DIV, B_MOD, B_POW, B_MAXIMUM, B_MINIMUM, B_EQ, B_NE, \
    B_LT, B_LE, B_GT, B_GE, B_LAND, B_LOR, B_LXOR = range(18)
R_SUM, R_PROD, R_MAX, R_MIN, R_ANY, R_ALL, R_ARGMAX, R_ARGMIN = range(8)
SCATTER_SET, SCATTER_ADD = 0, 1

_I64x8 = C.c_int64 * MAX_NDIM


class ArrayDesc(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("dtype", C.c_int32),
        ("ndim", C.c_int32),
        ("shape", _I64x8),
        ("strides", _I64x8),
        ("is_scalar", C.c_int32),
        ("_pad", C.c_int32),
        ("scalar_i", C.c_int64),
        ("scalar_f", C.c_double),
    ]


class IndexPlan(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32),
        ("n_idx", C.c_int32),
        ("shape", _I64x8),
        ("src_strides", _I64x8),
        ("idx_ptr", C.c_void_p * MAX_NDIM),
        ("idx_dtype", C.c_int32 * MAX_NDIM),
        ("idx_extent", _I64x8),
        ("idx_mult", _I64x8),
        ("idx_strides", _I64x8 * MAX_NDIM),
    ]


VM_MAX_INSTR, VM_MAX_LEAVES = 48, 8
VM_PUSH, VM_UNARY, VM_BINARY, VM_WHERE = range(4)
VM_SRC_STACK, VM_SRC_LEAF, VM_SRC_CONST = range(3)


def vm_ctrl(kind, op=0, ls=0, ll=0, rs=0, rl=0) -> int:
    """MDHIP_VM_CTRL of include/mdhip.h."""
    return kind | (op << 3) | (ls << 8) | (ll << 10) | (rs << 13) | (rl << 15)


class VmProgram(C.Structure):
    _fields_ = [
        ("n_instr", C.c_int32),
        ("n_leaves", C.c_int32),
        ("compute_dtype", C.c_int32),
        ("_pad", C.c_int32),
        ("ctrl", C.c_uint32 * VM_MAX_INSTR),
        ("imm", C.c_double * VM_MAX_INSTR),
        ("leaves", ArrayDesc * VM_MAX_LEAVES),
    ]


_EXC = {1: ValueError, 2: TypeError, 3: IndexError, 4: MemoryError, 5: RuntimeError}

_P = C.POINTER
_PROTOTYPES = {
    "mdhip_init": [C.c_int],
    "mdhip_device": [_P(C.c_int)],
    "mdhip_shutdown": [],
    "mdhip_debug_set_option": [C.c_char_p, C.c_int64],
    "mdhip_debug_get_option": [C.c_char_p, _P(C.c_int64)],
    "mdhip_alloc": [C.c_size_t, _P(C.c_void_p)],
    "mdhip_free": [C.c_void_p],
    "mdhip_empty_cache": [],
    "mdhip_mem_stats": [_P(C.c_int64)],
    "mdhip_host_alloc": [C.c_size_t, _P(C.c_void_p)],
    "mdhip_host_free": [C.c_void_p],
    "mdhip_h2d": [C.c_void_p, C.c_void_p, C.c_size_t],
    "mdhip_d2h": [C.c_void_p, C.c_void_p, C.c_size_t],
    "mdhip_d2d": [C.c_void_p, C.c_void_p, C.c_size_t],
    "mdhip_sync": [],
    "mdhip_random_fill": [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, _P(ArrayDesc)],
    "mdhip_random_permutation": [C.c_uint64, C.c_uint64, _P(ArrayDesc)],
    "mdhip_event_create": [_P(C.c_void_p)],
    "mdhip_event_record": [C.c_void_p],
    "mdhip_event_elapsed_ms": [C.c_void_p, C.c_void_p, _P(C.c_float)],
    "mdhip_event_destroy": [C.c_void_p],
    "mdhip_event_attach_next": [C.c_void_p, C.c_void_p],
    "mdhip_event_attach_cancel": [_P(C.c_int)],
    "mdhip_graph_begin": [],
    "mdhip_graph_end": [_P(C.c_void_p)],
    "mdhip_graph_launch": [C.c_void_p],
    "mdhip_graph_destroy": [C.c_void_p],
    "mdhip_unary": [C.c_int, _P(ArrayDesc), _P(ArrayDesc)],
    "mdhip_convert": [_P(ArrayDesc), _P(ArrayDesc)],
    "mdhip_binary": [C.c_int, _P(ArrayDesc), _P(ArrayDesc), _P(ArrayDesc), C.c_int],
    "mdhip_where": [_P(ArrayDesc), _P(ArrayDesc), _P(ArrayDesc), _P(ArrayDesc)],
    "mdhip_fill": [_P(ArrayDesc), _P(ArrayDesc)],
    "mdhip_arange": [_P(ArrayDesc), C.c_double, C.c_double],
    "mdhip_reduce": [C.c_int, _P(ArrayDesc), _P(ArrayDesc), C.c_uint32],
    "mdhip_var": [_P(ArrayDesc), _P(ArrayDesc), C.c_int32, C.c_int64, C.c_int],
    "mdhip_matmul": [_P(ArrayDesc), _P(ArrayDesc), _P(ArrayDesc)],
    "mdhip_matmul_bias_relu_sum": [_P(ArrayDesc), _P(ArrayDesc), _P(ArrayDesc), _P(ArrayDesc), _P(ArrayDesc)],
    "mdhip_gather": [_P(IndexPlan), C.c_void_p, C.c_int, _P(ArrayDesc)],
    "mdhip_scatter": [_P(IndexPlan), C.c_void_p, C.c_int, _P(ArrayDesc), C.c_int],
    "mdhip_nonzero_count": [_P(ArrayDesc), _P(C.c_int64)],
    "mdhip_nonzero_fill": [_P(ArrayDesc), C.c_int64, C.c_void_p],
    "mdhip_vm_eval": [_P(VmProgram), _P(ArrayDesc)],
    "mdhip_vm_reduce": [_P(VmProgram), C.c_int, _P(ArrayDesc), _P(ArrayDesc), C.c_uint32],
    "mdhip_vm_jit_probe": [_P(VmProgram), C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t],
    "mdhip_vm_jit_stats": [_P(C.c_int64)],
    "mdhip_vm_eval_multi": [_P(VmProgram), _P(ArrayDesc), C.c_int],
    "mdhip_vm_eval_reduce_cols": [_P(VmProgram), C.c_int, _P(ArrayDesc), _P(ArrayDesc)],
    "mdhip_vm_jit_probe_multi": [_P(VmProgram), C.c_int, C.c_char_p, C.c_size_t],
    "mdhip_comm_probe": [],
    "mdhip_comm_count": [_P(C.c_int)],
    "mdhip_comm_get_unique_id": [_P(C.c_uint8)],
    "mdhip_comm_init": [C.c_int, C.c_int, _P(C.c_uint8)],
    "mdhip_comm_allreduce_sum": [C.c_void_p, C.c_size_t, C.c_int],
    "mdhip_comm_allreduce_sum_async": [C.c_void_p, C.c_size_t, C.c_int],
    "mdhip_comm_wait": [],
    "mdhip_comm_destroy": [],
}
# every symbol include/mdhip.h declares (string-returning ones listed apart)
EXPORTED_SYMBOLS = sorted(list(_PROTOTYPES) + ["mdhip_target", "mdhip_last_error"])

PRODUCT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmdhip.so")
PRODUCT_TARGET = "hip:gfx950"


class Library:
    """A loaded libmdhip with checked calls: ``lib.call('mdhip_unary', ...)``."""

    def __init__(self, path: str):
        self.path = path
        self.cdll = C.CDLL(path, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
        for name, argtypes in _PROTOTYPES.items():
            fn = getattr(self.cdll, name)  # AttributeError = symbol missing: loud
            fn.argtypes = argtypes
            fn.restype = C.c_int
        self.cdll.mdhip_target.restype = C.c_char_p
        self.cdll.mdhip_target.argtypes = []
        self.cdll.mdhip_last_error.restype = C.c_char_p
        self.cdll.mdhip_last_error.argtypes = []
        self.target = self.cdll.mdhip_target().decode()
        self._initialised = False
        # bind hot entry points once (attribute lookups on CDLL are slow)
        trace = os.environ.get("MDHIP_TRACE")
        self._trace_file = open(trace, "a", buffering=1) if trace else None
        for name in _PROTOTYPES:
            fn = self._checked(getattr(self.cdll, name))
            if self._trace_file is not None:
                fn = self._traced(fn, name[len("mdhip_"):])
            setattr(self, name[len("mdhip_"):], fn)

    def _checked(self, fn):
        last_error = self.cdll.mdhip_last_error

        def call(*args):
            st = fn(*args)
            if st:
                raise _EXC.get(st, RuntimeError)(last_error().decode(errors="replace"))

        call.__name__ = fn.__name__
        return call

    def _traced(self, fn, name):
        """MDHIP_TRACE=<file>: one JSON line per C-ABI call — entry point, host-side duration, and for
        array arguments dtype code / shape / element strides (SURVEY.md §5 "call log at the C-ABI shim").
        Kernels are asynchronous: the duration is enqueue time, not execution time (use rocprofv3 for that)."""
        import json
        import time

        out = self._trace_file

        def call(*args):
            t0 = time.perf_counter()
            try:
                return fn(*args)
            finally:
                rec = {"call": name, "us": round((time.perf_counter() - t0) * 1e6, 1), "args": []}
                for a in args:
                    a = getattr(a, "_obj", a)  # byref(...) wrapper
                    if isinstance(a, ArrayDesc):
                        nd = a.ndim
                        rec["args"].append({"dtype": a.dtype, "shape": list(a.shape[:nd]), "strides": list(a.strides[:nd]),
                                            "scalar": bool(a.is_scalar)})
                    elif isinstance(a, (int, float)):
                        rec["args"].append(a)
                out.write(json.dumps(rec) + "\n")

        return call

    def ensure_init(self, device: int | None = None):
        if not self._initialised:
            if device is None:
                device = int(os.environ.get("MDHIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            self.init(device)
            self._initialised = True


_LIB: Library | None = None


_BIND_HOOKS: list = []   # callables(lib): told whenever the process is bound to a library (ndarray hands its entry points to _fastpath)


def _bound(lib: "Library") -> "Library":
    for hook in _BIND_HOOKS:
        hook(lib)
    return lib


def entry_address(lib: "Library", name: str) -> int:
    """Address of a C-ABI symbol of `lib` (for host code that calls it without ctypes: csrc/fastpath.c)."""
    return C.cast(getattr(lib.cdll, name), C.c_void_p).value


def load() -> Library:
    """Return the process-wide library, loading the PRODUCT build on first use."""
    global _LIB
    if _LIB is None:
        variant = os.environ.get("MDHIP_LIB_VARIANT")
        if variant:
            # EXPERIMENTS ONLY (scripts/gemm_ab.sh, prof_ab.sh, gemm_ablate.sh): an A/B build of the same gfx950 library
            # under scripts/ab/ (`make -C minidiff_amd/csrc variant NAME=..`) — named explicitly, announced on stderr, never
            # found by default. The product file itself is no longer swapped on the box by those scripts.
            import sys
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "ab", f"libmdhip_{variant}.so")
            if not os.path.exists(path):
                raise ImportError(f"MDHIP_LIB_VARIANT={variant}: {path} does not exist")
            print(f"[mdhip] EXPERIMENT BUILD in use: {path}", file=sys.stderr)
            lib = Library(path)
            if lib.target != PRODUCT_TARGET:
                raise ImportError(f"{path} reports target {lib.target!r}, expected {PRODUCT_TARGET!r}")
            _LIB = lib
            _LIB.ensure_init()
            return _bound(_LIB)
        if not os.path.exists(PRODUCT_LIB):
            raise ImportError(
                f"{PRODUCT_LIB} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C minidiff_amd/csrc`). "
                "minidiff_amd has no CPU fallback."
            )
        lib = Library(PRODUCT_LIB)
        if lib.target != PRODUCT_TARGET:
            raise ImportError(f"{PRODUCT_LIB} reports target {lib.target!r}, expected {PRODUCT_TARGET!r}")
        _LIB = lib
        _LIB.ensure_init()
        return _bound(_LIB)
    _LIB.ensure_init()
    return _LIB


def use_library(path: str, device: int | None = None) -> Library:
    """TESTS ONLY: bind the process to an explicitly named build of the C-ABI."""
    global _LIB
    _LIB = Library(path)
    _LIB.ensure_init(device)
    return _bound(_LIB)


def current() -> Library | None:
    return _LIB
