"""Opt-in lazy fusion of elementwise chains.

With ``MDHIP_LAZY=1`` (or ``ndarray.set_lazy(True)``) an elementwise backend
call does not launch: it returns a DeviceArray that carries an expression tree.
The tree grows while further elementwise calls consume it and is evaluated in
ONE pass by libmdhip's expression interpreter (``mdhip_vm_eval``) — or folded
into a reduction (``mdhip_vm_reduce``: full reduce, or the reduce-to-shape
column sum of the broadcast-gradient path) — when something needs the bytes:
a view, a matmul, a gather, an in-place write, a D2H copy. The tape above is
untouched; it simply sees arrays.

Default is eager (one kernel per backend call, the reference's execution model),
and bench.py reports the eager figure unless ``--lazy`` is given.

Bounds of a fused program (include/mdhip.h): 48 instructions, 8 distinct leaf
arrays, operand stack depth 4. Operators whose operand is a leaf array or a
constant carry it inside the instruction, so a chain costs one instruction per
operator. A tree that would exceed the bounds materialises its larger operand
first. Values are computed in ONE float type per program (float32 or float64 —
whatever NumPy's loop resolution gave each op); bool results travel as 0/1.
Integer loops are never fused.
"""
from __future__ import annotations

from . import _capi
from ._capi import VM_BINARY, VM_PUSH, VM_SRC_CONST, VM_SRC_LEAF, VM_SRC_STACK, VM_UNARY, VM_WHERE, vm_ctrl

LEAF, CONST, UNARY, BINARY, WHERE, GEMM = range(6)
MAX_INSTR, MAX_LEAVES, MAX_DEPTH = 44, 8, 4


class Expr:
    """Node of a pending expression. `cdt` is the program's float type code (F32/F64);
    `n` / `depth` are the instruction count and stack need of its emitted form."""

    __slots__ = ("kind", "code", "args", "n", "depth", "leaves", "cdt", "owner")

    def __init__(self, kind, code, args, n, depth, leaves, cdt):
        self.kind, self.code, self.args = kind, code, args
        self.n, self.depth, self.leaves, self.cdt = n, depth, leaves, cdt
        self.owner = None  # weakref to the pending DeviceArray this node is the value of (set by DeviceArray._pending)


def leaf(arr, cdt):
    return Expr(LEAF, 0, arr, 1, 1, {id(arr): arr}, cdt)


def const(value, cdt):
    return Expr(CONST, 0, float(value), 1, 1, {}, cdt)


def gemm(a, b, cdt):
    """A deferred matrix product a @ b (both concrete 2-D arrays). Not an interpreter instruction: the array that
    carries it materialises through the GEMM kernel, and enters elementwise programs as a LEAF — which lets the
    reduction of `where(a @ b + bias > 0, a @ b + bias, 0)` be recognised and run in the GEMM's epilogue."""
    return Expr(GEMM, 0, (a, b), 1, 1, {}, cdt)


def _simple(e) -> bool:
    return e.kind == LEAF or e.kind == CONST


def combine(kind, code, parts, cdt):
    leaves = {}
    for p in parts:
        if p.leaves:
            leaves.update(p.leaves)
    if kind == UNARY:
        (a,) = parts
        return Expr(kind, code, (a,), a.n + 1, a.depth, leaves, cdt)
    if kind == BINARY:
        a, b = parts
        sa, sb = _simple(a), _simple(b)
        if sa and sb:
            if a.kind == CONST and b.kind == CONST:      # two immediates: push one, fuse the other
                n, depth = 2, 1
            else:
                n, depth = 1, 1
        elif sb:
            n, depth = a.n + 1, a.depth
        elif sa:
            n, depth = b.n + 1, b.depth
        else:
            n, depth = a.n + b.n + 1, max(a.depth, b.depth + 1)
        return Expr(kind, code, (a, b), n, depth, leaves, cdt)
    c, a, b = parts  # WHERE: all three go through the stack
    n = c.n + a.n + b.n + 1
    depth = max(c.depth, a.depth + 1, b.depth + 2)
    return Expr(kind, code, (c, a, b), n, depth, leaves, cdt)


def fits(e: Expr) -> bool:
    return e.n <= MAX_INSTR and e.depth <= MAX_DEPTH and len(e.leaves) <= MAX_LEAVES


class _Emitter:
    """Postfix emission state. A plain object with methods — NOT nested recursive closures,
    which would form a reference cycle and keep the leaf arrays (HBM blocks) alive until
    Python's cyclic GC runs."""

    __slots__ = ("ctrl", "imm", "leaves", "leaf_ix")

    def __init__(self):
        self.ctrl, self.imm, self.leaves, self.leaf_ix = [], [], [], {}

    def src(self, x):
        """(source kind, leaf index, immediate) of a simple operand."""
        if x.kind == LEAF:
            k = id(x.args)
            ix = self.leaf_ix.get(k)
            if ix is None:
                ix = self.leaf_ix[k] = len(self.leaves)
                self.leaves.append(x.args)
            return VM_SRC_LEAF, ix, 0.0
        return VM_SRC_CONST, 0, x.args

    def put(self, word, value=0.0):
        self.ctrl.append(word)
        self.imm.append(value)

    def walk(self, x):
        if _simple(x):
            s, l, v = self.src(x)
            self.put(vm_ctrl(VM_PUSH, 0, 0, 0, s, l), v)
        elif x.kind == UNARY:
            self.walk(x.args[0])
            self.put(vm_ctrl(VM_UNARY, x.code))
        elif x.kind == BINARY:
            a, b = x.args
            sa, sb = _simple(a), _simple(b)
            if sa and sb and a.kind == CONST and b.kind == CONST:
                self.walk(a)
                s, l, v = self.src(b)
                self.put(vm_ctrl(VM_BINARY, x.code, VM_SRC_STACK, 0, s, l), v)
            elif sa and sb:
                s1, l1, v1 = self.src(a)
                s2, l2, v2 = self.src(b)
                self.put(vm_ctrl(VM_BINARY, x.code, s1, l1, s2, l2), v1 if s1 == VM_SRC_CONST else v2)
            elif sb:
                self.walk(a)
                s, l, v = self.src(b)
                self.put(vm_ctrl(VM_BINARY, x.code, VM_SRC_STACK, 0, s, l), v)
            elif sa:
                self.walk(b)
                s, l, v = self.src(a)
                self.put(vm_ctrl(VM_BINARY, x.code, s, l, VM_SRC_STACK, 0), v)
            else:
                self.walk(a)
                self.walk(b)
                self.put(vm_ctrl(VM_BINARY, x.code, VM_SRC_STACK, 0, VM_SRC_STACK, 0))
        else:
            for p in x.args:
                self.walk(p)
            self.put(vm_ctrl(VM_WHERE))


def emit(e: Expr):
    """-> (ctrl words, immediates, leaf arrays)."""
    em = _Emitter()
    em.walk(e)
    return em.ctrl, em.imm, em.leaves


def build_program(e: Expr, shape):
    ctrl, imm, leaves = emit(e)
    if len(ctrl) > _capi.VM_MAX_INSTR or len(leaves) > _capi.VM_MAX_LEAVES:
        raise ValueError("fused program exceeds the interpreter's limits")
    prog = _capi.VmProgram()
    prog.n_instr, prog.n_leaves, prog.compute_dtype = len(ctrl), len(leaves), e.cdt
    for i, (c, v) in enumerate(zip(ctrl, imm)):
        prog.ctrl[i] = c
        prog.imm[i] = v
    for i, arr in enumerate(leaves):
        prog.leaves[i] = arr.desc(shape)
    return prog, leaves
