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
The collective is ncclAllReduce(sum) issued by libmdhip on the same stream as
the kernels (RCCL over xGMI; `mdhip_comm_*` in include/mdhip.h), so it is
ordered after the last backward kernel without a host sync.

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

    def __init__(self, md, params, comm, force=False):
        self.md, self.params, self.comm, self.force = md, list(params), comm, force
        self.bucket = None
        self.nbytes = int(sum(p.size * np.dtype(p.dtype).itemsize for p in self.params))

    def __call__(self):
        B = self.md.backend
        if self.comm is None or (self.comm.world == 1 and not self.force):
            return
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("GradSync: a parameter has no gradient (was backward() run?)")
        if len(grads) == 1:
            raw = grads[0]._data
            if not _is_contiguous(raw):
                raw = B.copy(raw)
                self.params[0].grad = self.md.Tensor(raw)
            self.comm.allreduce_sum_(raw)
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
        self.comm.allreduce_sum_(self.bucket)


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
