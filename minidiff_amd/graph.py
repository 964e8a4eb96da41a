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
        self._lib.graph_begin()
        try:
            self.outputs = step()
        except BaseException:
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
