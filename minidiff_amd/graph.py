"""hipGraph capture / replay of a repeated sweep.

    sweep = CapturedSweep(step)      # runs `step()` twice: once eagerly (warms the allocator
                                     # cache and any run-time compiled kernels), once captured
    out = sweep.replay()             # one hipGraphLaunch; `out` is what the captured run returned

A replay re-executes every kernel of the captured sweep with the SAME device addresses,
so it overwrites the arrays the captured run produced (gradients included) — read them
after the replay, and feed new inputs by writing INTO the existing input arrays
(`x._data[...] = new_values`), not by creating new tensors. Everything inside the sweep
must be asynchronous: no `as_numpy()`, `item()`, printing, boolean-mask indexing or host
uploads (they need a device synchronisation and make the capture fail).
"""
from __future__ import annotations

import ctypes as C

from . import _capi


class CapturedSweep:
    def __init__(self, step, warmup: int = 1):
        self._lib = _capi.load() if _capi.current() is None else _capi.current()
        for _ in range(warmup):
            step()
        self._lib.sync()
        handle = C.c_void_p()
        # A capture records the kernels that are LAUNCHED between begin and end. In lazy mode a result still pending when the step
        # returns would be launched later, outside the graph, and every replay would miss it (a training loop replayed with
        # MDHIP_LAZY=1 drifted from its third sweep on): the recorded step runs eagerly; lazy mode comes back afterwards.
        from . import ndarray as _nd
        was_lazy = _nd.set_lazy(False)
        self._lib.graph_begin()
        try:
            self.outputs = step()
        except BaseException:
            _nd.set_lazy(was_lazy)
            try:
                self._lib.graph_end(C.byref(handle))
            except RuntimeError:
                pass
            if handle.value:  # the capture itself closed cleanly: give its graph and reserved blocks back
                try:
                    self._lib.graph_destroy(handle)
                except RuntimeError:
                    pass
            self._graph = None
            raise
        _nd.set_lazy(was_lazy)
        self._lib.graph_end(C.byref(handle))
        self._graph = handle
        self.replays = 0

    def replay(self):
        self._lib.graph_launch(self._graph)
        self.replays += 1
        return self.outputs

    def close(self):
        if self._graph is not None:
            self._lib.graph_destroy(self._graph)
            self._graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def can_capture(lib=None) -> bool:
    """Whether this build of the library can capture and instantiate a graph at all (an empty capture): False on the CPU
    test double. Cheap, issues nothing."""
    lib = lib or _capi.current() or _capi.load()
    probe = C.c_void_p()
    try:
        lib.sync()
        lib.graph_begin()
        lib.graph_end(C.byref(probe))
    except RuntimeError:
        return False
    if probe.value:
        lib.graph_destroy(probe)
    return True


class SegmentedSweep:
    """hipGraph replay of a data-parallel sweep (N > 1): the collectives must stay OUTSIDE the graphs (an RCCL call inside a
    stream capture would pull the communication stream and its events into the graph), so the sweep is captured as graph
    SEGMENTS cut at every communicator call:

        [graph 0: forward + backward up to the first weight-gradient panel] -> all-reduce(panel 0) (second stream)
        [graph 1: panel 1's GEMM] -> all-reduce(panel 1) ... -> wait

    The capturing run is a real run: each segment is launched as soon as it is closed and the communicator call is made
    eagerly, in order. `replay()` launches the segments and repeats the recorded communicator calls on the same buffers —
    no Python tape, no per-kernel dispatch: at 8 ranks the sweep's kernels are 65-250 us each and ~7 us of host dispatch
    per backend call would otherwise sit between them. `comm`: the communicator object the sweep's GradSync uses (its
    `allreduce_sum_async_`, `allreduce_sum_` and `wait` are intercepted during the capturing run only).
    Raises RuntimeError when the library cannot capture at all (the CPU test double) — before anything was issued."""

    def __init__(self, sweep, comm):
        self._lib = _capi.load() if _capi.current() is None else _capi.current()
        self._ops = []          # ("graph", handle) | ("call", bound function, args)
        if not can_capture(self._lib):
            raise RuntimeError("this build of the library cannot capture graphs")
        names = [n for n in ("allreduce_sum_async_", "allreduce_sum_", "wait") if hasattr(comm, n)]
        originals = {n: getattr(comm, n) for n in names}
        was_instance_attr = {n: n in vars(comm) for n in names}   # (a caller may have patched the INSTANCE already: tests count calls that way)

        def cut(fn):
            def wrapped(*a):
                self._close_segment()
                fn(*a)
                self._ops.append(("call", fn, a))   # (keeps the buffer views alive)
                self._lib.graph_begin()
            return wrapped

        for n in names:
            setattr(comm, n, cut(originals[n]))
        from . import ndarray as _nd
        was_lazy = _nd.set_lazy(False)     # (as in CapturedSweep: nothing may stay pending past the end of a segment)
        self._lib.graph_begin()
        try:
            self.outputs = sweep()
            self._close_segment()
        except BaseException:
            try:
                h = C.c_void_p()
                self._lib.graph_end(C.byref(h))
                if h.value:
                    self._lib.graph_destroy(h)
            except RuntimeError:
                pass
            self.close()
            raise
        finally:
            _nd.set_lazy(was_lazy)
            for n in names:   # put back exactly what was there: an instance attribute is restored, a class method is uncovered
                if was_instance_attr[n]:
                    setattr(comm, n, originals[n])
                else:
                    delattr(comm, n)
        self.replays = 0
        self.segments = sum(1 for op in self._ops if op[0] == "graph")
        self.calls = sum(1 for op in self._ops if op[0] == "call")

    def _close_segment(self):
        h = C.c_void_p()
        self._lib.graph_end(C.byref(h))
        self._lib.graph_launch(h)       # the capturing run executes for real, segment by segment
        self._ops.append(("graph", h))

    def replay(self):
        for op in self._ops:
            if op[0] == "graph":
                self._lib.graph_launch(op[1])
            else:
                op[1](*op[2])
        self.replays += 1
        return self.outputs

    def close(self):
        for op in self._ops:
            if op[0] == "graph" and op[1] is not None and op[1].value:
                try:
                    self._lib.graph_destroy(op[1])
                except RuntimeError:
                    pass
        self._ops = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SweepCache:
    """Replays a repeated sweep automatically: the device-side half of the reference's `reuse_graph`
    (minidiff/caching.py:14-65 memoises the TRAVERSAL of a graph it has seen; here the whole sweep's kernel
    sequence is memoised as a hipGraph), keyed by the tape's structural hash.

        cache = SweepCache(md)
        for batch in stream:
            x._data[...] = batch                  # new values go INTO the resident inputs
            out = cache.run(step)                 # 1st call: eager; 2nd: captured; then one hipGraphLaunch each

    `step()` records forward + backward on the tape (inside `md.reuse_graph()`, entered here), so every eager or
    capturing run yields the structural hash of its root (`md.last_root_hash`). A captured graph is kept under
    (step, hash); it is only replayed while the sweep keeps that structure:
      * a run whose hash differs from the previous one of the same `step` (another branch taken, another op) is
        simply a different entry: it runs eagerly, then captures its own graph;
      * a replay executes no Python, so it cannot see a change by itself: every `validate_every`-th call (default 64; 0 =
        never) of a captured entry runs eagerly instead and compares keys; on a mismatch the stale graph is destroyed.
    The key is the structural hash PLUS the tape's sweep signature (`tape._signature`: which inputs are the same tensor,
    shared intermediates, leaf shapes / dtypes, scalar constants, keyword arguments) — the structural hash alone maps every
    leaf and scalar to -1, so sum(sin(a)*sin(a)) and sum(sin(a)*sin(b)) would collide.
    A sweep that cannot be captured (a synchronising call inside it; the CPU test double) stays eager.
    `step` should RETURN the arrays it produces (e.g. the gradient tensors): a replay rewrites those very arrays,
    but it runs no Python, so attributes that the sweep rebinds (`x.grad = ...`) keep pointing at whatever the
    last Python-executed run left there."""

    def __init__(self, md, validate_every: int = 64):
        self.md = md
        self.validate_every = int(validate_every)
        self._entries = {}      # (id(step), hash) -> dict(seen, sweep, failed)
        self._current = {}      # id(step) -> hash of its latest eager/capturing run
        self._calls = {}        # id(step) -> calls since the last eager run
        self.stats = {"eager": 0, "captured": 0, "replayed": 0, "invalidated": 0, "uncapturable": 0}

    @staticmethod
    def _lib_sync():
        (_capi.current() or _capi.load()).sync()

    def _key(self):
        # the structural hash alone maps every leaf and scalar to -1 (as the reference's does): two sweeps that differ only in
        # WHICH inputs coincide, in shapes or in constants would share it, and a stale graph would replay silently
        return (self.md.last_root_hash, getattr(self.md, "last_root_signature", None))

    def _eager(self, step):
        with self.md.reuse_graph():   # (a fresh traversal table per Python-run sweep: the tape's memo is keyed by the structural hash alone)
            out = step()
            return out, self._key()

    def run(self, step):
        sid = id(step)
        h = self._current.get(sid)
        entry = self._entries.get((sid, h)) if h is not None else None
        if entry is not None and entry["sweep"] is not None:
            n = self._calls[sid] = self._calls.get(sid, 0) + 1
            if not (self.validate_every and n % self.validate_every == 0):
                self.stats["replayed"] += 1
                return entry["sweep"].replay()
            out, h2 = self._eager(step)     # validation run: same structure?
            self.stats["eager"] += 1
            if h2 != h:
                entry["sweep"].close()
                del self._entries[(sid, h)]
                self.stats["invalidated"] += 1
                self._current[sid] = h2
                self._entries.setdefault((sid, h2), {"seen": 1, "sweep": None, "failed": False})
            return out
        if entry is not None and entry["seen"] >= 1 and not entry["failed"]:
            # second sighting of this structure: capture it (the capture run executes the sweep once more)
            try:
                with self.md.reuse_graph():   # (a fresh traversal table per Python-run sweep: the tape's memo is keyed by the structural hash alone)
                    sweep = CapturedSweep(step, warmup=0)
                    h2 = self._key()
            except RuntimeError:
                entry["failed"] = True
                self.stats["uncapturable"] += 1
                out, h2 = self._eager(step)
                self.stats["eager"] += 1
                self._current[sid] = h2
                return out
            # (a capture only RECORDS the kernels: the sweep's results exist after the first launch of the graph)
            if h2 != h:   # the structure moved between two consecutive runs: do not keep this capture
                out = sweep.replay()
                self._lib_sync()
                sweep.close()
                self._current[sid] = h2
                self._entries.setdefault((sid, h2), {"seen": 1, "sweep": None, "failed": False})
                self.stats["eager"] += 1
                return out
            entry["sweep"] = sweep
            self._calls[sid] = 0
            self.stats["captured"] += 1
            return sweep.replay()
        out, h2 = self._eager(step)
        self.stats["eager"] += 1
        self._current[sid] = h2
        e = self._entries.setdefault((sid, h2), {"seen": 0, "sweep": None, "failed": False})
        e["seen"] += 1
        return out

    def close(self):
        for e in self._entries.values():
            if e["sweep"] is not None:
                e["sweep"].close()
        self._entries.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
