"""Data-parallel harness: batch rows sharded over ranks, ONE all-reduce of the
parameter gradients per sweep (no reference counterpart — the
reference has no distributed code).

One process per GPU. Each rank runs the unmodified tape on its shard with the
backend bound to device LOCAL_RANK; after backward() the parameter gradients
are summed across ranks in place on the device:
  * one parameter  -> its gradient buffer is all-reduced directly;
  * several        -> gradients are packed into one flat f32 bucket
                      ([W.grad || b.grad] = 16,781,312 floats for cfg4), one
                      collective, and the gradients become views of the bucket.
The collective is ncclAllReduce(sum) issued by libmdhip (RCCL over xGMI;
`mdhip_comm_*` in include/mdhip.h) without a host sync. With `overlap=True`
(default) the tape reports each parameter gradient the moment it is final
(`register_grad_ready_hook`; hooked inputs get their vjp first), and as soon as
the whole bucket is final the collective goes out on a second
stream while the compute stream carries on with the rest of backward (cfg2:
B.grad's all-reduce runs beside the A.grad GEMM); the weight gradient of a
matmul is produced in row panels inside the bucket, each panel's collective
leaving while the next panel's GEMM runs (cfg4, where that GEMM ends the sweep);
`GradSync.__call__` at the end of the sweep makes the compute stream wait. Without hooks firing (or
`overlap=False`) the collective is issued at that point on the compute stream.

`torch.distributed` is used for rendezvous / barriers / the ncclUniqueId
exchange only (gloo control plane). Communicators:
  RcclComm   product path (DeviceArray buffers, RCCL through the C-ABI)
  TorchComm  same data path through torch.distributed's nccl backend on a
             zero-copy view (fallback when the direct communicator cannot be
             created)
  HostComm   numpy buffers over gloo — CPU tests of the sharding logic only
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class RcclComm:
    def __init__(self, rank: int, world: int, dist=None):
        """Rendezvous that fails on EVERY rank or on none: whatever goes wrong on one rank (librccl cannot be
        loaded, ncclGetUniqueId / ncclCommInitRank fails) is agreed on over the control plane before anybody
        enters a call that would wait for the others, so callers can fall back together (bench.py)."""
        from . import _capi

        self.rank, self.world = rank, world
        self.lib = _capi.load()
        if world > 1 and dist is None:
            raise RuntimeError("a torch.distributed process group is needed to exchange the ncclUniqueId")
        uid = (C.c_uint8 * _capi.UID_BYTES)()
        err = None
        if rank == 0:
            try:
                self.lib.comm_get_unique_id(uid)  # (first RCCL call: this is where librccl is opened)
            except Exception as e:
                err = f"rank 0: {type(e).__name__}: {e}"
        if world > 1:
            box = [(err, bytes(uid))]
            dist.broadcast_object_list(box, src=0)  # rank 0 ALWAYS takes part, with the id or with its error
            err, raw = box[0]
            if err is not None:
                raise RuntimeError(f"RCCL rendezvous failed ({err})")
            uid = (C.c_uint8 * _capi.UID_BYTES).from_buffer_copy(raw)
        elif err is not None:
            raise RuntimeError(f"RCCL rendezvous failed ({err})")
        # ncclCommInitRank blocks until every rank has called it: a rank that cannot even start must say so first
        probe = getattr(self.lib, "comm_probe", None)
        my_err = None
        if probe is not None:
            try:
                probe()
            except Exception as e:
                my_err = f"rank {rank}: {type(e).__name__}: {e}"
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, my_err)
            bad = [e for e in errs if e]
            if bad:
                raise RuntimeError("RCCL rendezvous failed (" + "; ".join(bad) + ")")
        elif my_err:
            raise RuntimeError(f"RCCL rendezvous failed ({my_err})")
        self.lib.comm_init(world, rank, uid)

    def allreduce_sum_(self, arr):
        from . import ndarray as nd

        if not arr.is_c_contiguous:
            raise ValueError("allreduce needs a contiguous device buffer")
        nd.materialize(arr)
        nd._before_write(arr)
        self.lib.comm_allreduce_sum(arr.ptr, arr.size, nd.dtype_code(arr.dtype))

    def allreduce_sum_async_(self, arr):
        from . import ndarray as nd

        if not arr.is_c_contiguous:
            raise ValueError("allreduce needs a contiguous device buffer")
        nd.materialize(arr)
        nd._before_write(arr)
        self.lib.comm_allreduce_sum_async(arr.ptr, arr.size, nd.dtype_code(arr.dtype))

    def wait(self):
        self.lib.comm_wait()

    def close(self):
        self.lib.comm_destroy()


class TorchComm:
    """RCCL through torch.distributed (backend 'nccl') on a zero-copy view."""

    def __init__(self, rank: int, world: int, dist, torch):
        self.rank, self.world, self.dist, self.torch = rank, world, dist, torch
        self.group = dist.new_group(backend="nccl")

    def allreduce_sum_(self, arr):
        from . import _capi

        _capi.load().sync()  # our stream -> visible to torch's stream
        t = self.torch.as_tensor(arr, device=f"cuda:{self.torch.cuda.current_device()}")
        self.dist.all_reduce(t, group=self.group)
        self.torch.cuda.synchronize()

    def close(self):
        pass


class HostComm:
    """numpy over gloo: exercises sharding / bucketing on CPU (tests only)."""

    def __init__(self, rank: int, world: int, dist, torch):
        self.rank, self.world, self.dist, self.torch = rank, world, dist, torch

    def allreduce_sum_(self, arr):
        if not isinstance(arr, np.ndarray):
            # a DeviceArray of the CPU test double (its "device" memory is host memory): reduce it in place through a NumPy view
            from . import _capi

            lib = _capi.current()
            if lib is None or lib.target == _capi.PRODUCT_TARGET:
                raise TypeError("HostComm reduces host memory only (tests); use RcclComm for device arrays")
            if not arr.is_c_contiguous:
                raise ValueError("allreduce needs a contiguous buffer")
            lib.sync()
            raw = (C.c_char * arr.nbytes).from_address(arr.ptr)
            arr = np.frombuffer(raw, dtype=arr.dtype)
        t = self.torch.from_numpy(arr)
        self.dist.all_reduce(t)

    def close(self):
        pass


class GradSync:
    """Sums `.grad` of the given parameters across ranks after each backward().

    One flat bucket holds every parameter gradient; the gradients the sweep ends with are views of it.
    A parameter that is the right operand of a matmul (W in X @ W) gets its gradient X^T @ G computed in
    `panels` ROW PANELS written straight into the bucket: panel i's all-reduce leaves (second stream) while
    panel i+1's GEMM runs — on cfg4 the weight-gradient GEMM is the last kernel of the sweep, so without the
    panels the whole collective would be exposed. The rest of the bucket (biases, parameters whose gradient
    has several contributions) goes out in ONE collective when its last member is final."""

    def __init__(self, md, params, comm, force=False, overlap=True, panels=None):
        import os

        self.md, self.params, self.comm, self.force = md, list(params), comm, force
        self.bucket = None
        self.nbytes = int(sum(p.size * np.dtype(p.dtype).itemsize for p in self.params))
        self.active = not (comm is None or (comm.world == 1 and not force))
        self.overlap = bool(overlap and self.active and hasattr(md, "register_grad_ready_hook"))
        if panels is None:
            panels = int(os.environ.get("MDHIP_DP_PANELS", "4"))
        # a communicator without an asynchronous form would serialise every panel behind a host sync
        self.panels = max(1, int(panels)) if (self.overlap and hasattr(comm, "allreduce_sum_async_")) or isinstance(comm, HostComm) else 1
        if not self.overlap:
            self.panels = 1
        self._ready = set()
        self._paneled = set()   # ids of parameters whose gradient was produced (and sent) in panels this sweep
        self._in_flight = False
        self.overlapped = 0  # sweeps whose collective(s) went out from inside backward()
        self.panel_collectives = 0
        self._slots = None
        if self.overlap:
            for p in self.params:
                md.register_grad_ready_hook(p, self._on_ready, self._produce if self.panels > 1 else None)

    # ---- bucket layout: one slot per parameter, in the order given -------------------------------------
    def _ensure_bucket(self):
        if self.bucket is not None:
            return
        B = self.md.backend
        dt = self.params[0].dtype
        total = sum(p.size for p in self.params)
        self.bucket = B.zeros((total,), dtype=dt)
        self._slots, pos = {}, 0
        for p in self.params:
            self._slots[id(p)] = (pos, p.size)
            pos += p.size

    def _slot_view(self, p):
        pos, n = self._slots[id(p)]
        return self.bucket[pos:pos + n]

    # ---- gradient producer: weight gradient of a matmul, row panel by row panel --------------------------
    def _produce(self, node, index, grad):
        """Offered by the tape when `node.inputs[index]` (a hooked parameter) gets its single contribution.
        Handles `matmul(x, W)` with 2-D operands: W.grad = x^T @ grad (reference: minidiff/ops/definitions.py:487-492)
        — the same product, computed in row panels of W.grad that land in the bucket and are all-reduced at once."""
        if node.name != "matmul" or index != 1 or len(node.inputs) != 2:
            return None
        x, w = node.inputs
        if getattr(x, "ndim", 0) != 2 or w.ndim != 2 or grad.ndim != 2 or w.dtype != self.params[0].dtype or grad.dtype != w.dtype:
            return None
        if self._in_flight and not self._paneled:
            raise RuntimeError("GradSync: the previous sweep's collective was never joined (call the GradSync object after backward())")
        B = self.md.backend
        self._ensure_bucket()
        K, N = w.shape
        dest = B.reshape(self._slot_view(w), (K, N))
        xT = B.transpose(x._data if hasattr(x, "_data") else x)
        g = grad._data
        P = min(self.panels, max(1, K // 256))
        step = -(-K // P)
        step = -(-step // 256) * 256 if K >= 512 else step   # whole GEMM tiles per panel
        allreduce = getattr(self.comm, "allreduce_sum_async_", self.comm.allreduce_sum_)
        r0 = 0
        while r0 < K:
            r1 = min(K, r0 + step)
            B.matmul(xT[r0:r1], g, out=dest[r0:r1])
            allreduce(B.reshape(dest[r0:r1], ((r1 - r0) * N,)))
            self.panel_collectives += 1
            r0 = r1
        self._paneled.add(id(w))
        self._in_flight = True
        return self.md.Tensor(dest)

    def _on_ready(self, tensor):
        self._ready.add(id(tensor))
        if len(self._ready) == len(self.params):
            self._ready.clear()
            self._reduce(asynchronous=True)
            self._in_flight = True
            self.overlapped += 1

    def close(self):
        if self.overlap:
            for p in self.params:
                self.md.remove_grad_ready_hook(p)
            self.overlap = False

    def __call__(self):
        if not self.active:
            return
        self._ready.clear()
        if self._in_flight:  # issued from inside backward(): only join the streams
            self._in_flight = False
            self._paneled.clear()
            wait = getattr(self.comm, "wait", None)
            if wait is not None:
                wait()
            return
        self._reduce(asynchronous=False)
        self._paneled.clear()

    def _reduce(self, asynchronous):
        """Everything that has not gone out in panels: packed in the bucket, one collective per contiguous run."""
        B = self.md.backend
        allreduce = self.comm.allreduce_sum_
        if asynchronous:
            allreduce = getattr(self.comm, "allreduce_sum_async_", allreduce)
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("GradSync: a parameter has no gradient (was backward() run?)")
        if len(grads) == 1 and not self._paneled:
            raw = grads[0]._data
            if not _is_contiguous(raw):
                raw = B.copy(raw)
                self.params[0].grad = self.md.Tensor(raw)
            allreduce(raw)
            return
        self._ensure_bucket()
        dt = self.bucket.dtype
        run = None  # [start, end) of the current run of not-yet-sent slots
        runs = []
        for p, g in zip(self.params, grads):
            pos, n = self._slots[id(p)]
            if id(p) in self._paneled:
                if run is not None:
                    runs.append(run)
                    run = None
                continue
            view = self.bucket[pos:pos + n]
            view[...] = B.reshape(g._data if g.dtype == dt else B.astype(g._data, dt), (n,))
            p.grad = self.md.Tensor(B.reshape(view, g.shape))
            run = [pos, pos + n] if run is None else [run[0], pos + n]
        if run is not None:
            runs.append(run)
        if len(runs) == 1 and runs[0] == [0, self.bucket.size]:
            allreduce(self.bucket)
        else:
            for lo, hi in runs:
                allreduce(self.bucket[lo:hi])


def _is_contiguous(raw) -> bool:
    if isinstance(raw, np.ndarray):
        return raw.flags.c_contiguous
    return raw.is_c_contiguous


def shard_rows(n_rows: int, rank: int, world: int) -> slice:
    """Contiguous row block of rank `rank` (SURVEY.md §8e partitioning)."""
    if n_rows % world:
        raise ValueError(f"{n_rows} rows do not split evenly over {world} ranks")
    k = n_rows // world
    return slice(rank * k, (rank + 1) * k)
