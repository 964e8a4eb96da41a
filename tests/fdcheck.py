"""TEST INFRASTRUCTURE. Finite-difference gradient checker — this repo's counterpart of the
reference's test harness (reference: minidiff/utils.py:104-159
`calculate_finite_differences`, :163-197 `compute_grads`), written against the
tape API so it runs on any engine (HIP table on the GPU box, NumPy oracle).

Semantics kept: central differences (f(x+h) - f(x-h)) / 2h of a SCALAR `func`;
one perturbed copy of the input per element, stacked on a new leading axis
(`tile`), perturbed on the "diagonal" through an index-array getitem / in-place
add / setitem, and mapped with `vmap`; inputs that are not Tensors, do not
track gradients or are excluded yield None. Cost is O(n^2): tiny tensors only.
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np


def calculate_finite_differences(md, *input_tensors, func, h=1e-7, exclude=None):
    excluded = {id(x) for x in (exclude or [])}
    out = []
    with md.no_grad():
        for i, t in enumerate(input_tensors):
            if not isinstance(t, md.Tensor) or not t.allow_grad or id(t) in excluded:
                out.append(None)
                continue
            n, nd = t.size, t.ndim
            left, right = input_tensors[:i], input_tensors[i + 1:]

            def f(shifted, left=left, right=right):
                return func(*left, shifted, *right)

            mapped = md.vmap(f)
            coords = md.Tensor(np.array(tuple(np.ndindex(t.shape)), dtype=np.int64).reshape(n, nd))
            index = (md.arange(n), *[coords[:, d] for d in range(nd)])
            plus = md.tile(t.detach().copy(), (n,) + (1,) * nd)
            minus = md.tile(t.detach().copy(), (n,) + (1,) * nd)
            plus[index] += h
            minus[index] -= h
            out.append(((mapped(plus) - mapped(minus)) / (2 * h)).reshape(t.shape))
    return out


def compute_grads(md, *input_tensors, func, h=1e-7, exclude=None):
    """-> (finite-difference grads, autodiff grads) on detached copies of the inputs."""
    excluded = {id(x) for x in (exclude or [])}
    copies, copied_exclude = [], []
    for t in input_tensors:
        c = t.copy().detach(allow_grad=True) if isinstance(t, md.Tensor) else deepcopy(t)
        copies.append(c)
        if id(t) in excluded:
            copied_exclude.append(c)
    result = func(*copies)
    result.backward(retain_grads=True)
    auto = [c.grad if isinstance(c, md.Tensor) else None for c in copies]
    manual = calculate_finite_differences(md, *copies, func=func, h=h, exclude=copied_exclude)
    return manual, auto
