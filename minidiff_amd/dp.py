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
B.grad's all-reduce runs beside the A.grad GEMM); `GradSync.__call__` at the end
of the sweep makes the compute stream wait for it. Without hooks firing (or
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
        from . import _capi

        self.rank, self.world = rank, world
        self.lib = _capi.load()
        uid = (C.c_uint8 * _capi.UID_BYTES)()
        if rank == 0:
            self.lib.comm_get_unique_id(uid)
        if world > 1:
            if dist is None:
                raise RuntimeError("a torch.distributed process group is needed to exchange the ncclUniqueId")
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * _capi.UID_BYTES).from_buffer_copy(box[0])
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

    def allreduce_sum_(self, arr: np.ndarray):
        t = self.torch.from_numpy(arr)
        self.dist.all_reduce(t)

    def close(self):
        pass


class GradSync:
    """Sums `.grad` of the given parameters across ranks after each backward()."""

    def __init__(self, md, params, comm, force=False, overlap=True):
        self.md, self.params, self.comm, self.force = md, list(params), comm, force
        self.bucket = None
        self.nbytes = int(sum(p.size * np.dtype(p.dtype).itemsize for p in self.params))
        self.active = not (comm is None or (comm.world == 1 and not force))
        self.overlap = bool(overlap and self.active and hasattr(md, "register_grad_ready_hook"))
        self._ready = set()
        self._in_flight = False
        self.overlapped = 0  # sweeps whose collective went out from inside backward()
        if self.overlap:
            for p in self.params:
                md.register_grad_ready_hook(p, self._on_ready)

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
            wait = getattr(self.comm, "wait", None)
            if wait is not None:
                wait()
            return
        self._reduce(asynchronous=False)

    def _reduce(self, asynchronous):
        B = self.md.backend
        allreduce = self.comm.allreduce_sum_
        if asynchronous:
            allreduce = getattr(self.comm, "allreduce_sum_async_", allreduce)
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("GradSync: a parameter has no gradient (was backward() run?)")
        if len(grads) == 1:
            raw = grads[0]._data
            if not _is_contiguous(raw):
                raw = B.copy(raw)
                self.params[0].grad = self.md.Tensor(raw)
            allreduce(raw)
            return
        dt = grads[0].dtype
        total = sum(g.size for g in grads)
        if self.bucket is None or self.bucket.size != total:
            self.bucket = B.zeros((total,), dtype=dt)
        pos = 0
        for p, g in zip(self.params, grads):
            n = g.size
            view = self.bucket[pos:pos + n]
            view[...] = B.reshape(g._data if g.dtype == dt else B.astype(g._data, dt), (n,))
            p.grad = self.md.Tensor(B.reshape(view, g.shape))
            pos += n
        allreduce(self.bucket)


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
