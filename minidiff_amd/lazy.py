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

Bounds of a fused program (include/mdhip.h): 48 postfix instructions, 8 distinct
leaf arrays, 16 constants, operand stack depth 4. A tree that would exceed them
materialises its larger operand first. Values are computed in ONE float type per
program (float32 or float64 — whatever NumPy's loop resolution gave each op);
bool results travel as 0/1. Integer loops are never fused.
"""
from __future__ import annotations

from . import _capi

LEAF, CONST, UNARY, BINARY, WHERE = 0, 1, 2, 3, 4
MAX_INSTR, MAX_LEAVES, MAX_CONSTS, MAX_DEPTH = 40, 8, 16, 4


class Expr:
    """Node of a pending expression. `cdt` is the program's float type code (F32/F64)."""

    __slots__ = ("kind", "code", "args", "n", "depth", "leaves", "cdt")

    def __init__(self, kind, code, args, n, depth, leaves, cdt):
        self.kind, self.code, self.args = kind, code, args
        self.n, self.depth, self.leaves, self.cdt = n, depth, leaves, cdt


def leaf(arr, cdt):
    return Expr(LEAF, 0, arr, 1, 1, {id(arr): arr}, cdt)


def const(value, cdt):
    return Expr(CONST, 0, float(value), 1, 1, {}, cdt)


def combine(kind, code, parts, cdt):
    """Postfix cost of evaluating `parts` left to right then applying the operator."""
    n = 1
    depth = 0
    leaves = {}
    for i, p in enumerate(parts):
        n += p.n
        d = p.depth + i
        if d > depth:
            depth = d
        if p.leaves:
            leaves.update(p.leaves)
    return Expr(kind, code, tuple(parts), n, depth, leaves, cdt)


def fits(e: Expr) -> bool:
    return e.n <= MAX_INSTR and e.depth <= MAX_DEPTH and len(e.leaves) <= MAX_LEAVES


def emit(e: Expr):
    """-> (kinds, args, leaf arrays, consts)."""
    kinds, args, leaves, consts = [], [], [], []
    leaf_ix, const_ix = {}, {}

    def walk(x):
        if x.kind == LEAF:
            k = id(x.args)
            if k not in leaf_ix:
                leaf_ix[k] = len(leaves)
                leaves.append(x.args)
            kinds.append(LEAF)
            args.append(leaf_ix[k])
        elif x.kind == CONST:
            if x.args not in const_ix:
                const_ix[x.args] = len(consts)
                consts.append(x.args)
            kinds.append(CONST)
            args.append(const_ix[x.args])
        else:
            for p in x.args:
                walk(p)
            kinds.append(x.kind)
            args.append(x.code)

    walk(e)
    return kinds, args, leaves, consts


def build_program(e: Expr, shape):
    kinds, args, leaves, consts = emit(e)
    if len(kinds) > _capi.VM_MAX_INSTR or len(leaves) > _capi.VM_MAX_LEAVES or len(consts) > _capi.VM_MAX_CONSTS:
        raise ValueError("fused program exceeds the interpreter's limits")
    prog = _capi.VmProgram()
    prog.n_instr, prog.n_leaves, prog.n_consts, prog.compute_dtype = len(kinds), len(leaves), len(consts), e.cdt
    for i, (k, a) in enumerate(zip(kinds, args)):
        prog.kind[i] = k
        prog.arg[i] = a
    for i, c in enumerate(consts):
        prog.consts[i] = c
    for i, arr in enumerate(leaves):
        prog.leaves[i] = arr.desc(shape)
    return prog, leaves
