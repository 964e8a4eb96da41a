"""Eager reverse-mode tape that drives a backend function table.

Why this exists: the kernels are reached only through the backend boundary, and
on the GPU box the reference's own Python (Tensor / create_op_func / OpNode)
is not available. This module is the host-side mirror of that caller layer —
same public names, same argument meaning, same error behaviour — so parity
tests read like the reference's tests and, crucially, so the *sequence of
backend calls* per graph is the one the reference issues (pinned
by tests/golden traces captured from the real reference).

Mirrored interfaces (reference file:line):
  Tensor, grad-mode context managers ........ minidiff/tensor.py:19-69, 92-433
  creation / index helpers ................... minidiff/tensor.py:453-677
  op construction (forward + vjp closures) ... minidiff/ops/wrapping.py:117-178
  op table and gradient formulas ............. minidiff/ops/definitions.py:15-559
  node, toposort, backward ................... minidiff/topology.py:15-200
The graph-reuse cache (minidiff/caching.py) is not mirrored (out of scope).

`build_engine(table)` returns a namespace bound to ONE backend table: the HIP
table for the product, or the NumPy table under oracle/ when tests and the CPU
baseline need the reference's arithmetic. The tape itself does no arithmetic.
"""
from __future__ import annotations

import types

import numpy as np
from contextvars import ContextVar
from math import prod as _pyprod


def _visit(node, seen, out):
    if node is None:
        return
    for t in node.tensor_inputs:
        if id(t) in seen:
            continue
        seen.add(id(t))
        _visit(t.op_node, seen, out)
        out.append(t)


def _visit_paths(node, prefix, seen, out):
    """Like _visit, but records for every tensor the input-index walk that reaches it from the root node."""
    if node is None:
        return
    for i, t in enumerate(node.inputs):
        if not hasattr(t, "op_node") or id(t) in seen:
            continue
        seen.add(id(t))
        path = prefix + (i,)
        _visit_paths(t.op_node, path, seen, out)
        out.append(path)


def _signature(root):
    """What a REPLAY of a recorded sweep depends on and the structural hash leaves out (it maps every leaf and every
    non-tensor to -1, as the reference's topology.py:52-63 does): which inputs are the SAME tensor (sin(a)*sin(a) vs
    sin(a)*sin(b)), which intermediates are shared, leaf shapes and dtypes, scalar constants and keyword arguments. One int
    per graph; every node is hashed once (graph.SweepCache keys captured hipGraphs by (structural hash, this))."""
    leaf_index, memo = {}, {}

    def value_sig(x):
        """A non-op-input value (a constant operand, a getitem key, a keyword argument). Array-likes are NEVER repr()-ed: the
        repr of a device array is a device-to-host copy and a stream sync — one per such op in every backward(), and inside a
        stream capture it makes the sweep uncapturable — and NumPy's repr elides the middle of a large array, so two index
        arrays could collide. They are identified like leaves: type, shape, dtype and first-sighting index by identity."""
        if hasattr(x, "op_node"):                     # a Tensor inside a key / kwarg (x[(idx_tensor, slice(None))])
            if x.op_node is None or getattr(x, "is_leaf", False):
                return ("L", leaf_index.setdefault(id(x), len(leaf_index)), tuple(x.shape), str(x.dtype))
            return node_sig(x.op_node)
        if isinstance(x, (tuple, list)):
            return (type(x).__name__,) + tuple(value_sig(v) for v in x)
        if isinstance(x, dict):
            return ("dict",) + tuple((k, value_sig(v)) for k, v in sorted(x.items(), key=lambda kv: repr(kv[0])))
        if isinstance(x, slice):
            return ("slice", value_sig(x.start), value_sig(x.stop), value_sig(x.step))
        if isinstance(x, np.ndarray) and x.size <= 64:    # a small HOST constant: by value (a replay bakes its upload in)
            return ("H", tuple(x.shape), str(x.dtype), x.tobytes())
        if hasattr(x, "shape") and hasattr(x, "dtype") and not isinstance(x, (np.generic,)):   # DeviceArray / large ndarray
            return ("A", type(x).__name__, leaf_index.setdefault(id(x), len(leaf_index)), tuple(x.shape), str(x.dtype))
        return ("C", type(x).__name__, repr(x))       # Python / NumPy scalars, None, Ellipsis, strings, dtypes

    def node_sig(node):
        got = memo.get(id(node))
        if got is not None:
            return ("R", got[1])                      # a shared intermediate: by its first-visit index
        index = len(memo)
        memo[id(node)] = (None, index)
        parts = [node.name, tuple(sorted((k, value_sig(v)) for k, v in node.kwargs.items()))]
        for x in node.inputs:
            parts.append(value_sig(x))
        h = hash(tuple(parts))
        memo[id(node)] = (h, index)
        return ("N", h)

    return hash(node_sig(root))


def build_engine(B, name: str = "engine"):
    """Create the md-like namespace (Tensor, ops, helpers) over backend table `B`."""
    E = types.SimpleNamespace()
    E.backend = B
    E.__name__ = name

    grad_on = ContextVar(f"{name}_grad_on", default=True)
    new_grads_on = ContextVar(f"{name}_new_grads_on", default=True)

    # ------------------------------------------------------------ grad modes ----
    class _Mode:
        def __init__(self, value):
            self.value = value

        def __enter__(self):
            self.saved = grad_on.get()
            grad_on.set(self.value)

        def __exit__(self, *exc):
            grad_on.set(self.saved)

    class no_grad(_Mode):
        def __init__(self):
            super().__init__(False)

    class enable_grad(_Mode):
        def __init__(self, enable):
            super().__init__(enable)

    class disable_new_grads:
        def __enter__(self):
            self.saved = (grad_on.get(), new_grads_on.get())
            grad_on.set(False)
            new_grads_on.set(False)

        def __exit__(self, *exc):
            grad_on.set(self.saved[0])
            new_grads_on.set(self.saved[1])

    # ----------------------------------------------- reuse_graph (tape memoisation) ----
    # Counterpart of the reference's minidiff/caching.py:14-65: inside `with md.reuse_graph():` every op node
    # carries a STRUCTURAL id (the ids of its tensor inputs' producers, -1 for leaves and non-tensors, and the
    # op's name: topology.py:46-74), the root's hash keys a cache of the backward traversal — stored as
    # input-index walks from the root, so a later graph of the same structure reuses it without another
    # topological sort (topology.py:152-162). `E.last_root_hash` exposes the hash of the latest backward()
    # root: minidiff_amd.graph.SweepCache keys captured hipGraphs by it.
    caching_on = ContextVar(f"{name}_caching_graph", default=False)
    cached_paths = ContextVar(f"{name}_cached_indices", default=None)
    E.last_root_hash = None

    E.last_root_signature = None

    class reuse_graph:
        def __init__(self, table=None):
            self.table = table   # a dict to keep the memoised traversals in ACROSS scopes (graph.SweepCache); None: a fresh one

        def __enter__(self):
            self.prev = (caching_on.get(), cached_paths.get())
            caching_on.set(True)
            cached_paths.set({} if self.table is None else self.table)
            return self

        def __exit__(self, *exc):
            caching_on.set(self.prev[0])
            cached_paths.set(self.prev[1])

    E.reuse_graph = reuse_graph
    E.currently_caching = caching_on.get

    def backward_paths_for_root(root):
        if not caching_on.get():
            raise ValueError("Not currently preserving graph")
        table = cached_paths.get()
        h = root.hash
        paths = table.get(h)
        if paths is None:
            paths = []
            _visit_paths(root, (), set(), paths)
            paths = table[h] = tuple(paths)
        return paths

    E.backward_paths_for_root = backward_paths_for_root

    E.no_grad, E.enable_grad, E.disable_new_grads = no_grad, enable_grad, disable_new_grads
    E.grad_allowed_ = grad_on.get
    E.set_allow_grad = grad_on.set
    E.new_grads_allowed_ = new_grads_on.get
    E.set_allow_new_grads = new_grads_on.set

    def try_unwrap(t):
        if isinstance(t, Tensor):
            return t._data
        if isinstance(t, tuple):
            return tuple(try_unwrap(x) for x in t)
        if isinstance(t, list):
            return [try_unwrap(x) for x in t]
        if isinstance(t, dict):
            return {k: try_unwrap(v) for k, v in t.items()}
        return t

    E.try_unwrap = try_unwrap
    _PLAIN = (int, float, bool, type(None))

    # ----------------------------------------------------------------- node ----
    class Node:
        """One recorded op: inputs + one vjp closure per input."""

        __slots__ = ("vjps", "inputs", "kwargs", "name", "pass_kwargs", "tensor_inputs", "op_ids", "_hash")

        def __init__(self, vjps, inputs, kwargs, name, pass_kwargs):
            self.vjps = vjps
            self.inputs = inputs
            self.kwargs = kwargs or {}
            self.name = name or ""
            self.pass_kwargs = pass_kwargs
            tin = []
            for x in inputs:
                if isinstance(x, Tensor):
                    tin.append(x)
                    x.graph_refs += 1
            self.tensor_inputs = tin
            self.op_ids = None
            if caching_on.get():  # structural id (topology.py:52-63)
                # (a producer enters by its HASH, not by its nested id tuple: tuple hashes are not cached, and hashing nested
                # tuples re-walks a shared node once per path to it — exponential on graphs that reuse intermediates)
                ids = []
                for x in inputs:
                    if not isinstance(x, Tensor) or x.is_leaf or x.op_node is None or x.op_node.op_ids is None:
                        ids.append(-1)
                    else:
                        ids.append(x.op_node.hash)
                ids.append(self.name)
                self.op_ids = tuple(ids)
                self._hash = hash(self.op_ids)

        @property
        def hash(self):
            return self._hash if self.op_ids is not None else hash(None)

        def push(self, grad, pending=None):
            """Chain rule for this node: evaluate each vjp, undo broadcasting, accumulate.
            `pending` (only when gradient-ready hooks are registered, see
            `register_grad_ready_hook`) counts the contributions each hooked leaf still
            expects: hooked inputs are served first and their hook fires the moment the
            last contribution has been accumulated, so a collective on that gradient
            overlaps the vjps that remain."""
            inputs, vjps = self.inputs, self.vjps
            order = range(len(inputs))
            if pending:
                order = sorted(order, key=lambda i: id(inputs[i]) not in pending)  # stable: hooked first
            kw = self.kwargs if self.pass_kwargs else None
            for i in order:
                inp, vjp = inputs[i], vjps[i]
                if vjp is None or not isinstance(inp, Tensor) or not inp._allow_grad:
                    continue
                g = None
                if pending and pending.get(id(inp)) == 1 and inp.grad is None and not E.grad_allowed_():
                    # the ONLY contribution to a hooked leaf, first order: its hook may produce the gradient itself
                    # (dp.GradSync: the weight gradient of a matmul in row panels, straight into the all-reduce bucket)
                    producer = E.grad_ready_hooks[id(inp)][2]
                    if producer is not None:
                        g = producer(self, i, grad)
                if g is None:
                    g = vjp(*inputs, grad, **kw) if kw else vjp(*inputs, grad)
                if g.shape != inp.shape:
                    g = E.unbroadcast(g, inp.shape)
                inp.grad = g if inp.grad is None else inp.grad + g
                if pending and id(inp) in pending:
                    pending[id(inp)] -= 1
                    if pending[id(inp)] == 0:
                        del pending[id(inp)]
                        E.grad_ready_hooks[id(inp)][1](inp)

        def backward(self, seed, retain_grads=False, cleanup_mode="prune", allow_higher_order=False, reset_grads=True):
            if cleanup_mode not in ("keep", "prune", "destroy"):
                raise ValueError(f"Cleanup mode not recognized ({cleanup_mode})")
            if allow_higher_order:
                retain_grads = True
                if cleanup_mode == "destroy":
                    cleanup_mode = "prune"
            if caching_on.get() and self.op_ids is not None:
                path = []
                for walk in backward_paths_for_root(self):   # memoised order, re-bound to THIS graph's tensors
                    node = self
                    for i in walk:
                        t = node.inputs[i]
                        node = t.op_node
                    path.append(t)
                E.last_root_hash = self.hash
                E.last_root_signature = _signature(self)
            else:
                path = _toposort(self)
            if reset_grads:
                for t in path:
                    t.grad = None
            pending = None
            if E.grad_ready_hooks:
                pending = {}
                for node in [self] + [t.op_node for t in path if not t.is_leaf]:
                    for inp, vjp in zip(node.inputs, node.vjps):
                        if vjp is not None and isinstance(inp, Tensor) and inp.allow_grad and id(inp) in E.grad_ready_hooks:
                            pending[id(inp)] = pending.get(id(inp), 0) + 1
            with enable_grad(allow_higher_order):
                self.push(seed, pending)
                for t in reversed(path):
                    if t.is_leaf:
                        continue
                    node = t.op_node
                    node.push(t.grad, pending)
                    if not retain_grads:
                        t.grad = None
                    if cleanup_mode == "keep":
                        continue
                    if cleanup_mode == "destroy":
                        t.wipe()
                        continue
                    if t.graph_refs > 0:
                        continue
                    for child in node.tensor_inputs:
                        child.graph_refs -= 1
                    t.wipe()

        def __repr__(self):
            return f"{self.name}({', '.join(str(x) for x in self.inputs)})"

    def _toposort(root):
        """Post-order DFS over tensor inputs (inputs before their consumers), the
        reference's visiting order (topology.py:106-128). Deliberately not a
        self-referencing closure: that would form a reference cycle holding every
        intermediate (hundreds of MB of HBM each) until Python's cyclic GC runs."""
        seen, out = set(), []
        _visit(root, seen, out)
        return out

    E.OpNode = Node

    # gradient-ready hooks (no reference counterpart: the data-parallel harness).
    # id(tensor) -> (tensor, fn); fn(tensor) runs inside backward() as soon as tensor.grad is final
    E.grad_ready_hooks = {}

    def register_grad_ready_hook(tensor, fn, producer=None):
        """`fn(tensor)` runs the moment `tensor.grad` is final inside backward(). `producer(node, input_index, grad)`
        (optional) is offered the computation of that gradient when it has a single contribution: it returns the
        gradient Tensor, or None to leave it to the op's own vjp."""
        E.grad_ready_hooks[id(tensor)] = (tensor, fn, producer)

    def remove_grad_ready_hook(tensor):
        E.grad_ready_hooks.pop(id(tensor), None)

    E.register_grad_ready_hook = register_grad_ready_hook
    E.remove_grad_ready_hook = remove_grad_ready_hook

    # --------------------------------------------------------------- tensor ----
    class Tensor:
        def __init__(self, data, allow_grad=False, dtype=None):
            data = try_unwrap(data)
            if data is None:
                data = B.tensor_constructor([])
            if not isinstance(data, B.tensor_class):
                data = B.tensor_constructor(data)
            if dtype is not None:
                data = data.astype(dtype)
            self._data = data
            self._allow_grad = allow_grad
            self._iterator = None
            self.graph_refs = 0
            self.grad = None
            self.op_node = None

        @property
        def graphed(self):
            return self.graph_refs > 0 or self.op_node is not None

        @property
        def is_leaf(self):
            return self.op_node is None

        @property
        def allow_grad(self):
            return self._allow_grad

        @allow_grad.setter
        def allow_grad(self, allow):
            if not allow and not self.is_leaf:
                raise ValueError(
                    "Turning off gradient tracking for intermediate tensors will almost always break chain rule in backprop")
            if self._allow_grad == allow:
                return
            self.grad = None
            self._allow_grad = allow

        T = property(lambda self: E.transpose(self))
        shape = property(lambda self, _f=B.tensor_shape: _f(self._data))
        size = property(lambda self: B.tensor_size(self._data))
        ndim = property(lambda self: B.tensor_ndim(self._data))
        dtype = property(lambda self: B.tensor_dtype(self._data))

        def as_numpy(self):
            return B.as_numpy(self._data)

        def backward(self, retain_grads=False, cleanup_mode="prune", allow_higher_order=False, reset_grads=True):
            if not self._allow_grad or self.is_leaf:
                return
            self.grad = E.ones_like(self, allow_grad=allow_higher_order)
            self.op_node.backward(self.grad, retain_grads=retain_grads, cleanup_mode=cleanup_mode,
                                  allow_higher_order=allow_higher_order, reset_grads=reset_grads)

        def wipe(self):
            self.op_node = None

        def detach(self, allow_grad=False):
            return Tensor(self._data, allow_grad=allow_grad)

        def item(self):
            if self.size != 1:
                raise ValueError("Only Tensors with a single element can be reduced to a Python scalar")
            return B.tensor_item(self._data)

        # method spellings of ops
        def ravel(self, order="C"): return E.ravel(self, order=order)
        def flatten(self, order="C"): return E.flatten(self, order=order)
        def astype(self, dtype): return E.astype(self, dtype)
        def transpose(self, axes=None): return E.transpose(self, axes=axes)
        def sum(self, axis=None, keepdims=False): return E.sum(self, axis=axis, keepdims=keepdims)
        def copy(self): return E.copy(self)
        def clip(self, a_min=None, a_max=None): return E.clip(self, a_min=a_min, a_max=a_max)
        def reshape(self, shape): return E.reshape(self, shape)
        def dot(self, other): return E.dot(self, other)
        def matmul(self, other): return E.matmul(self, other)
        def add(self, other): return E.add(self, other)
        def multiply(self, other): return E.multiply(self, other)

        def _validate_mutation(self):
            if self._allow_grad and grad_on.get() and self.graphed:
                raise ValueError("In-place operations can break computation graphs during backprop")

        def __mod__(self, o): return E.mod(self, o)
        def __matmul__(self, o): return E.matmul(self, o)
        def __add__(self, o): return E.add(self, o)
        def __radd__(self, o): return E.add(o, self)
        def __sub__(self, o): return E.subtract(self, o)
        def __rsub__(self, o): return E.subtract(o, self)
        def __mul__(self, o): return E.multiply(self, o)
        def __rmul__(self, o): return E.multiply(o, self)
        def __truediv__(self, o): return E.true_divide(self, o)
        def __rtruediv__(self, o): return E.true_divide(o, self)
        def __floordiv__(self, o): return E.floor_divide(self, o)
        def __rfloordiv__(self, o): return E.floor_divide(o, self)
        def __pow__(self, o): return E.power(self, o)
        def __rpow__(self, o): return E.power(o, self)
        def __neg__(self): return -1 * self

        def __imod__(self, o):
            self._validate_mutation(); self._data %= try_unwrap(o); return self

        def __imatmul__(self, o):
            self._validate_mutation(); self._data @= o._data; return self

        def __iadd__(self, o):
            self._validate_mutation(); self._data += try_unwrap(o); return self

        def __isub__(self, o):
            self._validate_mutation(); self._data -= try_unwrap(o); return self

        def __imul__(self, o):
            self._validate_mutation(); self._data *= try_unwrap(o); return self

        def __itruediv__(self, o):
            self._validate_mutation(); self._data /= try_unwrap(o); return self

        def __ifloordiv__(self, o):
            self._validate_mutation(); self._data //= try_unwrap(o); return self

        def __ipow__(self, o):
            self._validate_mutation(); self._data **= try_unwrap(o); return self

        def __repr__(self): return B.repr(self._data)
        def __len__(self): return B.len(self._data)
        def __getitem__(self, key): return E.getitem(self, key)

        def __setitem__(self, key, val):
            self._validate_mutation()
            self._data[try_unwrap(key)] = try_unwrap(val)

        def __gt__(self, v): return E.greater(self, v)
        def __ge__(self, v): return E.greater_equal(self, v)
        def __lt__(self, v): return E.less(self, v)
        def __le__(self, v): return E.less_equal(self, v)
        def __eq__(self, v): return E.equal(self, v)
        def __ne__(self, v): return E.not_equal(self, v)
        def __and__(self, v): return E.logical_and(self, v)
        def __or__(self, v): return E.logical_or(self, v)
        def __xor__(self, v): return E.logical_xor(self, v)
        def __invert__(self): return E.invert(self)
        __hash__ = object.__hash__

        def __iter__(self):
            n = B.tensor_size(self._data)
            length = len(self) if n > 1 else n
            return (self[i] for i in range(length))

        @property
        def __array_interface__(self):
            return B.array_interface(self._data)

        def __array__(self, dtype=None, copy=None):
            return B.array(self._data, dtype=dtype, copy=copy)

    E.Tensor = Tensor
    _tensor_class = B.tensor_class

    def _wrap(data, allow):
        """Tensor(data, allow_grad=allow) for a value that IS a backend array (an op's result): the fields, without the
        constructor's unwrapping and conversion."""
        t = Tensor.__new__(Tensor)
        t._data = data
        t._allow_grad = allow
        t._iterator = None
        t.graph_refs = 0
        t.grad = None
        t.op_node = None
        return t

    # ------------------------------------------------------- op construction ----
    def _wants_grad(inputs):
        if not grad_on.get():
            return False
        for x in inputs:
            if isinstance(x, Tensor) and x.allow_grad:
                return True
        return False

    def _check_inputs(inputs, tensor_only):
        ok = False
        for t in inputs:
            is_t = isinstance(t, Tensor)
            ok = is_t
            if (is_t and not tensor_only) or (not is_t and tensor_only):
                break
        if ok:
            return
        if tensor_only:
            raise ValueError("This function only supports minidiff Tensors")
        raise ValueError("This function requires at least one minidiff Tensor argument")

    def lift(fn):
        """backend function -> Tensor function (unwrap, call, wrap)."""
        def lifted(*args, **kwargs):
            allow = _wants_grad(args)
            # (the common call — Tensors and scalars, no keyword arguments — without the generic recursion)
            raw = [a._data if type(a) is Tensor else (a if type(a) in _PLAIN else try_unwrap(a)) for a in args]
            out = fn(*raw, **try_unwrap(kwargs)) if kwargs else fn(*raw)
            return _wrap(out, allow) if type(out) is _tensor_class else Tensor(out, allow_grad=allow)
        lifted.__name__ = getattr(fn, "__name__", "backend_fn")
        lifted.backend_fn = fn
        return lifted

    E.as_minidiff = lift

    def make_op(forward, vjps, pass_kwargs=False, differentiable=True, tensor_only=False, name=None):
        if not differentiable:
            vjps = [None] * len(vjps)
        opname = name or forward.__name__
        raw_fn = getattr(forward, "backend_fn", None)

        def lifted_op(*inputs, **kwargs):
            """`op` below for a forward that is a lifted backend function: input check, gradient mode, unwrapping and
            wrapping in one pass over the arguments (same checks, same errors, same backend call)."""
            allow, n_t, raw = False, 0, []
            for a in inputs:
                if type(a) is Tensor or isinstance(a, Tensor):
                    n_t += 1
                    if a._allow_grad:
                        allow = True
                    raw.append(a._data)
                elif type(a) in _PLAIN:
                    raw.append(a)
                else:
                    raw.append(try_unwrap(a))
            if tensor_only:
                if n_t != len(inputs) or n_t == 0:
                    raise ValueError("This function only supports minidiff Tensors")
            elif n_t == 0:
                raise ValueError("This function requires at least one minidiff Tensor argument")
            if allow and not grad_on.get():
                allow = False
            data = raw_fn(*raw, **try_unwrap(kwargs)) if kwargs else raw_fn(*raw)
            out = _wrap(data, allow) if type(data) is _tensor_class else Tensor(data, allow_grad=allow)
            if differentiable and allow:
                out.op_node = Node(vjps, inputs, kwargs, opname, pass_kwargs)
            return out

        def op(*inputs, **kwargs):
            _check_inputs(inputs, tensor_only)
            allow = _wants_grad(inputs)
            out = forward(*inputs, **kwargs)
            if out.op_node is not None:
                out = out.detach()
            out.allow_grad = allow
            if differentiable and allow and grad_on.get():
                out.op_node = Node(vjps, inputs, kwargs, opname, pass_kwargs)
            return out

        if raw_fn is not None:
            op = lifted_op
        op.__name__ = opname
        op.__qualname__ = f"<op func '{opname}'>"
        return op

    E.create_op_func = lambda forward_func, grad_funcs, propagate_kwargs=False, is_differentiable=True, tensor_only=False, op_name=None: \
        make_op(forward_func, grad_funcs, propagate_kwargs, is_differentiable, tensor_only, op_name)

    def unary(fn_name, vjp=None, **kw):
        kw["tensor_only"] = True
        return make_op(lift(getattr(B, fn_name)), [vjp], name=fn_name, **kw)

    def binary(fn_name, vx=None, vy=None, **kw):
        return make_op(lift(getattr(B, fn_name)), [vx, vy], name=kw.pop("name", fn_name), **kw)

    def ternary(fn_name, vx=None, vy=None, vz=None, **kw):
        return make_op(lift(getattr(B, fn_name)), [vx, vy, vz], name=fn_name, **kw)

    # ----------------------------------------------------- non-diff helpers ----
    def _creator(fn_name, unwrap_first=True):
        fn = getattr(B, fn_name)

        def create(*args, allow_grad=False, **kwargs):
            return Tensor(fn(*try_unwrap(args), **try_unwrap(kwargs)), allow_grad=allow_grad)
        create.__name__ = fn_name
        return create

    for _n in ("ones_like", "ones", "zeros_like", "zeros", "full_like", "concatenate", "unravel_index", "repeat",
               "tile", "arange", "stack", "load", "rand", "randn", "binomial", "permutation", "take_along_axis"):
        setattr(E, _n, _creator(_n))

    def full(shape, allow_grad=False):  # the reference drops the fill value (tensor.py:480-481)
        return Tensor(B.full(shape), allow_grad=allow_grad)

    def randint(low, high=None, size=None, allow_grad=False):
        return Tensor(B.randint(try_unwrap(low), high=try_unwrap(high), size=size), allow_grad=allow_grad)

    def choice(a, size=None, replace=True, p=None):
        return Tensor(B.choice(try_unwrap(a), size=size, replace=replace, p=try_unwrap(p)))

    def index_add(a, indices, b=None):
        B.index_add(try_unwrap(a), try_unwrap(indices), try_unwrap(b))

    def put_along_axis(arr, indices, values, axis):
        B.put_along_axis(arr._data, indices._data, try_unwrap(values), axis)

    def isin(element, test_elements):
        return B.isin(try_unwrap(element), try_unwrap(test_elements))

    def save(file, arr):
        B.save(file, arr._data)

    def shuffle(x):
        B.shuffle(x._data)

    def split(ary, indices_or_sections, axis=0, allow_grad=False):
        parts = B.split(ary._data, try_unwrap(indices_or_sections), axis=axis)
        return [Tensor(p, allow_grad=allow_grad) for p in parts]

    def vmap(fun):
        def backend_func(arr, *args, **kwargs):
            return fun(Tensor(arr), *[Tensor(x) for x in args], **{k: Tensor(v) for k, v in kwargs.items()})._data
        mapped = B.vmap(backend_func)

        def wrapper(*args, **kwargs):
            return Tensor(mapped(*try_unwrap(args), **try_unwrap(kwargs)))
        return wrapper

    E.full, E.randint, E.choice, E.index_add, E.put_along_axis = full, randint, choice, index_add, put_along_axis
    E.isin, E.save, E.shuffle, E.split, E.vmap = isin, save, shuffle, split, vmap
    for _dt in ("float64", "float32", "float16", "uint64", "uint32", "uint16", "uint8", "int64", "int32", "int16", "int8", "bool"):
        setattr(E, _dt, getattr(B, _dt))
    E.newaxis = None

    # ------------------------------------------------------------- gradients ----
    # Formulas are those of minidiff/ops/definitions.py (cited per op); operand
    # order inside each product is kept so the backend sees the same calls.
    def g_squeeze(a, grad, axis=None, **_):  # definitions.py:15-25
        if axis is None:
            axis = [i for i, n in enumerate(a.shape) if n == 1]
        if not axis:
            return grad
        return E.expand_dims(grad, axis)

    def _tensordot_axes(x, y, axes):
        if isinstance(axes, int):
            axes = (tuple(range(x.ndim - axes, x.ndim)), tuple(range(axes)))
        free_x = tuple(i for i in range(x.ndim) if i not in axes[0])
        free_y = tuple(i for i in range(y.ndim) if i not in axes[1])
        return axes, free_x, free_y

    def g_tensordot_x(x, y, grad, axes=2):  # definitions.py:28-61
        axes, free_x, free_y = _tensordot_axes(x, y, axes)
        g_axes = tuple(range(grad.ndim - len(free_y), grad.ndim))
        res = E.tensordot(grad, y, axes=(g_axes, free_y))
        perm = [0] * x.ndim
        nf = len(free_x)
        fi = ci = 0
        for i in range(x.ndim):
            if i < nf:
                perm[free_x[fi]] = i
                fi += 1
            else:
                perm[axes[0][ci]] = i
                ci += 1
        return E.transpose(res, axes=perm)

    def g_tensordot_y(x, y, grad, axes=2):  # definitions.py:64-95
        axes, free_x, free_y = _tensordot_axes(x, y, axes)
        g_axes = tuple(range(len(free_x)))
        res = E.tensordot(x, grad, axes=(free_x, g_axes))
        nc = len(axes[0])
        perm = [0] * y.ndim
        ci = fi = 0
        for i in range(y.ndim):
            if i < nc:
                perm[axes[1][ci]] = i
                ci += 1
            else:
                perm[free_y[fi]] = i
                fi += 1
        return E.transpose(res, axes=perm)

    def g_max(x, grad, axis=None, **_):  # definitions.py:98-114
        if axis is None:
            index = E.argmax(x, axis=axis, keepdims=True)
            return grad[index]
        if not axis:
            return grad
        idx = E.argmax(x, axis=axis, keepdims=True)
        grad = grad.reshape(idx.shape)
        ret = E.zeros_like(x)
        E.put_along_axis(ret, idx, grad, axis=axis)
        return ret

    def g_min(x, grad, axis=None, **_):  # definitions.py:117-127
        idx = E.argmin(x, axis=axis, keepdims=True)
        grad = grad.reshape(idx.shape)
        ret = E.zeros_like(x)
        E.put_along_axis(ret, idx, grad, axis=axis)
        return ret

    def g_prod(x, grad, axis=None, **_):  # definitions.py:130-141
        if axis == ():
            return grad.reshape(x.shape)
        p = E.prod(x, axis=axis, keepdims=True)
        grad = grad.reshape(p.shape)
        return E.where(x == 0, 0, grad * p / x)

    def g_transpose(x, grad, axes=None):  # definitions.py:144-152
        if axes is None:
            return E.transpose(grad)
        back = [-1] * len(axes)
        for i, d in enumerate(axes):
            back[d.item()] = i
        return E.transpose(grad, axes=back)

    def unbroadcast_forward(x, target_shape):  # definitions.py:157-183
        if x.shape == target_shape:
            return x
        lead = tuple(range(x.ndim - len(target_shape)))
        if len(lead) != 0:
            x = x.sum(axis=lead)
        nd = min(len(target_shape), x.ndim)
        stretched = tuple(i for i in range(nd) if x.shape[i] > 1 and target_shape[i] == 1)
        if len(stretched) != 0:
            x = x.sum(axis=stretched, keepdims=True)
        if x.size == _pyprod(target_shape):
            return x.reshape(target_shape)
        return E.broadcast_to(x, target_shape)

    def g_getitem(x, key, grad):  # definitions.py:186-189
        ret = E.zeros_like(x)
        E.index_add(ret, key, grad)
        return ret

    def g_sum(x, grad, axis=None, **_):  # definitions.py:224-262
        if isinstance(axis, int):
            axis = tuple(axis)
        if axis is None:
            return grad
        if not axis:
            return grad
        shape = x.shape
        nd = len(shape)
        summed = [i for i in range(nd) if i in axis]
        dims = [shape[i] for i in summed]
        ns = len(summed)
        tiled = E.tile(grad, dims + [1] * (nd - ns))
        perm = [0] * nd
        shifted = 0
        for i in reversed(range(nd)):
            if shifted != ns and i == summed[-(shifted + 1)]:
                perm[i] = ns - 1 - shifted
                shifted += 1
            else:
                perm[i] = i + shifted
        return E.transpose(tiled, axes=perm)

    def g_mean(x, grad, axis=None, **_):  # definitions.py:192-206
        if axis is None:
            return grad / x.size
        if not axis:
            return grad
        if isinstance(axis, int):
            return grad / x.shape[axis]
        dims = Tensor([x.shape[d] for d in axis])
        return g_sum(x, grad, axis=axis) / E.prod(dims)

    def g_std(x, grad, axis=None, **kwargs):  # definitions.py:209-221
        if axis is None:
            axis = E.arange(x.ndim)
        if not axis:
            return E.zeros_like(x)
        mu = E.mean(x, axis=axis)
        n = _pyprod([d for i, d in enumerate(x.shape) if i in axis])
        return grad * (x - mu) / (E.std(x, axis=axis, **kwargs) * n)

    # --------------------------------------------------------------- op table ----
    # unary (definitions.py:266-420)
    E.absolute = unary("absolute", lambda x, grad: grad * E.sign(x))
    E.abs = E.absolute
    for _n in ("all", "any", "argmax", "argmin", "argwhere", "ceil", "floor", "invert", "logical_not", "sign"):
        setattr(E, _n, unary(_n, differentiable=False))
    for _n in ("atleast_1d", "atleast_2d", "atleast_3d", "copy"):
        setattr(E, _n, unary(_n, lambda x, grad: grad))
    E.cos = unary("cos", lambda x, grad: grad * -E.sin(x))
    E.cosh = unary("cosh", lambda x, grad: grad * E.sinh(x))
    E.exp = unary("exp", lambda x, grad: grad * E.exp(x))
    E.flatten = unary("flatten", lambda x, grad, order="C": E.reshape(grad, x.shape, order=order))
    E.flip = unary("flip", lambda x, grad, **kw: E.flip(grad, **kw), pass_kwargs=True)
    E.log = unary("log", lambda x, grad: grad / x)
    E.max = unary("max", g_max, pass_kwargs=True)
    E.mean = unary("mean", g_mean, pass_kwargs=True)
    E.min = unary("min", g_min, pass_kwargs=True)
    E.prod = unary("prod", g_prod, pass_kwargs=True)
    E.ravel = unary("ravel", lambda x, grad, order="C": E.reshape(grad, x.shape, order=order))
    E.sin = unary("sin", lambda x, grad: grad * E.cos(x))
    E.sinh = unary("sinh", lambda x, grad: grad * E.cosh(x))
    E.sqrt = lambda a, **kw: E.power(a, 0.5, **kw)
    E.square = lambda a, **kw: E.power(a, 2, **kw)
    E.squeeze = unary("squeeze", g_squeeze)
    E.std = unary("std", g_std, pass_kwargs=True)
    E.sum = unary("sum", g_sum, pass_kwargs=True)
    E.tan = unary("tan", lambda x, grad: grad * (1 / E.cos(x) ** 2))
    E.tanh = unary("tanh", lambda x, grad: grad * (1 / E.cosh(x) ** 2))
    E.transpose = unary("transpose", g_transpose, pass_kwargs=True)

    # binary (definitions.py:424-536)
    E.add = binary("add", lambda x, y, grad: grad, lambda x, y, grad: grad)
    E.astype = binary("astype", lambda x, dtype, grad: grad.astype(x.dtype))
    E.broadcast_to = binary("broadcast_to", lambda x, shape, grad: E.unbroadcast(grad, x.shape))
    E.dot = binary("dot", lambda x, y, grad: grad * y, lambda x, y, grad: grad * x)
    E.expand_dims = binary("expand_dims", lambda x, axis, grad: E.squeeze(grad, axis=axis))
    for _n in ("equal", "floor_divide", "greater", "greater_equal", "less", "less_equal", "logical_and", "logical_or",
               "logical_xor", "not_equal"):
        setattr(E, _n, binary(_n, differentiable=False))
    E.getitem = binary("getitem", g_getitem, name="index")
    E.matmul = binary("matmul", lambda x, y, grad: E.matmul(grad, y.T), lambda x, y, grad: E.matmul(x.T, grad),
                      tensor_only=True)
    E.mod = binary("mod", lambda x, y, grad: E.where(x % y == 0, 0, grad), lambda x, y, grad: E.where(x % y == 0, 0, grad))
    E.multiply = binary("multiply", lambda x, y, grad: grad * y, lambda x, y, grad: grad * x)
    E.power = binary("power", lambda x, y, grad: grad * y * (x ** (y - 1)), lambda x, y, grad: grad * E.log(x) * x ** y)
    E.reshape = binary("reshape", lambda x, y, grad: grad.reshape(x.shape))
    E.subtract = binary("subtract", lambda x, y, grad: grad, lambda x, y, grad: -grad)
    E.tensordot = binary("tensordot", g_tensordot_x, g_tensordot_y, tensor_only=True, pass_kwargs=True)
    E.true_divide = binary("true_divide", lambda x, y, grad: grad / y, lambda x, y, grad: grad * (-x / y ** 2))
    E.unbroadcast = make_op(unbroadcast_forward, [lambda x, shape, grad: E.broadcast_to(grad, x.shape), None],
                            name="unbroadcast_forward")

    # ternary (definitions.py:538-559)
    E.clip = ternary("clip", lambda x, a_min, a_max, grad: grad * E.logical_and(
        1 if a_min is None else x > a_min, 1 if a_max is None else x < a_max))
    E.swapaxes = ternary("swapaxes", lambda x, a1, a2, grad, **kw: E.swapaxes(grad, a1, a2, **kw), pass_kwargs=True)
    E.where = ternary("where", None, lambda c, y, z, grad: grad * c, lambda c, y, z, grad: grad * (1 - c))

    E.OP_NAMES = [
        "absolute", "abs", "all", "any", "argmax", "argmin", "argwhere", "atleast_1d", "atleast_2d", "atleast_3d", "ceil",
        "copy", "cos", "cosh", "exp", "flatten", "flip", "floor", "invert", "log", "logical_not", "max", "min", "mean", "prod",
        "ravel", "sign", "sin", "sinh", "sqrt", "square", "squeeze", "std", "sum", "tan", "tanh", "transpose", "add", "astype",
        "broadcast_to", "dot", "equal", "expand_dims", "floor_divide", "getitem", "greater", "greater_equal", "less",
        "less_equal", "logical_and", "logical_or", "logical_xor", "matmul", "mod", "multiply", "not_equal", "power",
        "reshape", "subtract", "tensordot", "true_divide", "unbroadcast", "clip", "swapaxes", "where",
    ]
    return E


_HIP_ENGINE = None


def hip_engine():
    """The product engine: tape bound to the MI355X backend table (loads libmdhip)."""
    global _HIP_ENGINE
    if _HIP_ENGINE is None:
        from . import _capi
        from .hip_backend import HipBackendTable

        if _capi.current() is None:
            _capi.load()
        _HIP_ENGINE = build_engine(HipBackendTable, "hip")
    return _HIP_ENGINE
