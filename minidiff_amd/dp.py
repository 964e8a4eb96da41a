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
            if not arr.is_c_contiguous:
                raise ValueError("allreduce needs a contiguous buffer")
            if lib is None or lib.target == _capi.PRODUCT_TARGET:
                # a REHEARSAL on real device memory (scripts/rehearse_two_ranks_one_gpu.sh: two ranks sharing one GPU, where RCCL refuses
                # to build a communicator): staged through the host — copy out, gloo, copy back. Never a measured path.
                from . import ndarray as nd

                host = arr.get()
                self.dist.all_reduce(self.torch.from_numpy(host))
                arr[...] = nd.asarray(host)
                return
            lib.sync()
            raw = (C.c_char * arr.nbytes).from_address(arr.ptr)
            arr = np.frombuffer(raw, dtype=arr.dtype)
        t = self.torch.from_numpy(arr)
        self.dist.all_reduce(t)

    def close(self):
        pass


def auto_panels(K: int, N: int, cap: int = 8) -> int:
    """Row panels for a K x N weight gradient: as many as leave every panel's GEMM a FULL round of 128 x 128 output tiles
    on the 256 CUs (a panel below one round runs at the small-grid rate and gains no overlap worth it); cfg4's 4096 x 4096
    weight -> 4 panels of 1024 rows (256 tiles each), a 1024 x 1024 layer -> 1 (the literal single all-reduce)."""
    tiles = -(-K // 128) * -(-N // 128)
    return max(1, min(cap, tiles // 256, max(1, K // 256)))


class GradSync:
    """Sums `.grad` of the given parameters across ranks after each backward().

    One flat bucket holds every parameter gradient, in the order the parameters were given; the gradients the sweep ends
    with are views of it. A parameter that is the right operand of a matmul (W in X @ W) gets its gradient X^T @ G computed
    in ROW PANELS written straight into the bucket: panel i's all-reduce leaves (second stream) while panel i+1's GEMM
    runs — on cfg4 the weight-gradient GEMM is the last kernel of the sweep, so un-panelled the whole collective would be
    exposed. Bucket neighbours whose gradient is already final (cfg4: the bias, bucket = [W || b]) ride on the adjacent
    panel's collective, so the sweep issues `panels` collectives and no 16-KiB one of pure latency. Whatever is left goes
    out in ONE collective per contiguous run when the last parameter is final, or — parameters the sweep never reported —
    when the object is called at the end of the sweep (which raises if such a parameter has no gradient at all).
    `panels`: None = by shape (`auto_panels`; MDHIP_DP_PANELS overrides), 1 = the single all-reduce of the whole bucket."""

    def __init__(self, md, params, comm, force=False, overlap=True, panels=None):
        import os

        self.md, self.params, self.comm, self.force = md, list(params), comm, force
        self.bucket = None
        self.nbytes = int(sum(p.size * np.dtype(p.dtype).itemsize for p in self.params))
        self.active = not (comm is None or (comm.world == 1 and not force))
        self.overlap = bool(overlap and self.active and hasattr(md, "register_grad_ready_hook"))
        if panels is None and os.environ.get("MDHIP_DP_PANELS"):
            panels = int(os.environ["MDHIP_DP_PANELS"])
        # a communicator without an asynchronous form would serialise every panel behind a host sync
        can_panel = self.overlap and (hasattr(comm, "allreduce_sum_async_") or isinstance(comm, HostComm))
        self._panels_req = (max(1, int(panels)) if panels is not None else None) if can_panel else 1
        self.panels = self._panels_req or 0   # what the last panelled gradient used (0: none produced yet, count by shape)
        self._ready = set()     # ids of parameters reported final in this sweep
        self._sent = set()      # ids of parameters whose bucket slot has been handed to a collective in this sweep
        self._in_flight = False
        self.overlapped = 0  # sweeps whose collective(s) all went out from inside backward()
        self.panel_collectives = 0
        self._slots = None
        if self.overlap:
            for p in self.params:
                md.register_grad_ready_hook(p, self._on_ready, self._produce if self._panels_req != 1 else None)

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

    def _pack(self, p):
        """Copy p's (final) gradient into its bucket slot and make the gradient a view of the slot."""
        B = self.md.backend
        g = p.grad
        pos, n = self._slots[id(p)]
        view = self.bucket[pos:pos + n]
        view[...] = B.reshape(g._data if g.dtype == self.bucket.dtype else B.astype(g._data, self.bucket.dtype), (n,))
        p.grad = self.md.Tensor(B.reshape(view, g.shape))

    def _ride_along(self, w, lo, hi, first, last):
        """Extend the bucket range [lo, hi) of w's first / last panel over neighbouring slots whose gradients are final
        already and not yet sent (they are packed now): one collective instead of two."""
        k = next(i for i, p in enumerate(self.params) if p is w)
        if last:
            for p in self.params[k + 1:]:
                if id(p) not in self._ready or id(p) in self._sent or p.grad is None:
                    break
                self._pack(p)
                self._sent.add(id(p))
                hi = self._slots[id(p)][0] + self._slots[id(p)][1]
        if first:
            for p in reversed(self.params[:k]):
                if id(p) not in self._ready or id(p) in self._sent or p.grad is None:
                    break
                self._pack(p)
                self._sent.add(id(p))
                lo = self._slots[id(p)][0]
        return lo, hi

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
        if id(w) in self._sent or id(w) in self._ready:
            raise RuntimeError("GradSync: a second backward() reached a parameter whose gradient of THIS sweep has already been "
                               "handed to the all-reduce (call the GradSync object after every backward())")
        B = self.md.backend
        self._ensure_bucket()
        K, N = w.shape
        P = self._panels_req if self._panels_req is not None else auto_panels(K, N)
        P = min(P, max(1, K // 256))
        self.panels = P
        pos_w = self._slots[id(w)][0]
        dest = B.reshape(self._slot_view(w), (K, N))
        xT = B.transpose(x._data if hasattr(x, "_data") else x)
        g = grad._data
        step = -(-K // P)
        step = -(-step // 256) * 256 if K >= 512 else step   # whole GEMM tiles per panel
        allreduce = getattr(self.comm, "allreduce_sum_async_", self.comm.allreduce_sum_)
        self._sent.add(id(w))
        r0 = 0
        while r0 < K:
            r1 = min(K, r0 + step)
            B.matmul(xT[r0:r1], g, out=dest[r0:r1])
            lo, hi = self._ride_along(w, pos_w + r0 * N, pos_w + r1 * N, first=r0 == 0, last=r1 == K)
            allreduce(self.bucket[lo:hi])
            self.panel_collectives += 1
            r0 = r1
        self._in_flight = True
        return self.md.Tensor(dest)

    def _on_ready(self, tensor):
        if id(tensor) in self._ready:
            raise RuntimeError("GradSync: a parameter's gradient was reported final twice before the sweep's collectives were "
                               "joined (two backward() calls without calling the GradSync object in between)")
        self._ready.add(id(tensor))
        if len(self._ready) == len(self.params):
            self._reduce_rest(asynchronous=True)
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
        try:
            if len(self._sent) < len(self.params):
                # parameters the sweep never reported final (no hooks, or a parameter the graph did not use): reduce them now,
                # on the compute stream (a synchronous all-reduce joins the in-flight ones first: one issue order per communicator)
                self._reduce_rest(asynchronous=False)
            if self._in_flight:
                wait = getattr(self.comm, "wait", None)
                if wait is not None:
                    wait()
        finally:
            self._in_flight = False
            self._ready.clear()
            self._sent.clear()

    def _reduce_rest(self, asynchronous):
        """Everything that has not gone out yet: packed in the bucket, one collective per contiguous run of slots."""
        B = self.md.backend
        allreduce = self.comm.allreduce_sum_
        if asynchronous:
            allreduce = getattr(self.comm, "allreduce_sum_async_", allreduce)
        rest = [p for p in self.params if id(p) not in self._sent]
        if not rest:
            return
        if any(p.grad is None for p in rest):
            raise RuntimeError("GradSync: a parameter has no gradient (was backward() run? is every parameter used by the sweep?)")
        if len(self.params) == 1 and self.bucket is None:
            raw = rest[0].grad._data
            if not _is_contiguous(raw):
                raw = B.copy(raw, order="C")
                self.params[0].grad = self.md.Tensor(raw)
            self._sent.add(id(rest[0]))
            allreduce(raw)
            return
        self._ensure_bucket()
        run = None  # [start, end) of the current run of not-yet-sent slots
        runs = []
        for p in self.params:
            pos, n = self._slots[id(p)]
            if id(p) in self._sent:
                if run is not None:
                    runs.append(run)
                    run = None
                continue
            self._pack(p)
            self._sent.add(id(p))
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
