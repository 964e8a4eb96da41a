"""One rank of the world_size-2 gloo rehearsal of the data-parallel sweep
(launched by tests/test_dp_gloo.py). Runs cfg4 and cfg2 at reduced size on the
NumPy oracle table, shards the batch rows, all-reduces parameter grads through
minidiff_amd.dp.GradSync, and checks them against the single-process full-batch
gradients computed locally."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from minidiff_amd import dp, workloads  # noqa: E402
from minidiff_amd.tape import build_engine  # noqa: E402
from oracle.numpy_table import NumpyOracleTable  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    md = build_engine(NumpyOracleTable, "oracle")
    comm = dp.HostComm(rank, world, dist, torch)

    # cfg4: global batch split by rows, bucketed [W.grad || b.grad] all-reduce
    full_state, full_step = workloads.make_cfg4(md, batch=64, d_in=24, d_out=40, rank=0, world=1)
    full_step()
    st, step = workloads.make_cfg4(md, batch=64, d_in=24, d_out=40, rank=rank, world=world)
    assert st["X"].shape == (64 // world, 24)
    assert np.array_equal(st["X"].as_numpy(), full_state["X"].as_numpy()[dp.shard_rows(64, rank, world)])
    sync = dp.GradSync(md, st["params"], comm)
    assert sync.nbytes == (24 * 40 + 40) * 4
    step()
    assert sync.overlapped == 1  # the bucket went out from inside backward(), when its last member became final
    sync()
    for name in ("W", "b"):
        got, exp = st[name].grad.as_numpy(), full_state[name].grad.as_numpy()
        assert got.shape == exp.shape and got.dtype == exp.dtype
        err = np.abs(got - exp).max() / np.abs(exp).max()
        assert err < 1e-5, (name, err)
    assert sync.bucket is not None and sync.bucket.size == 24 * 40 + 40
    # second sweep reuses the bucket and must give the same answer (no accumulation across sweeps)
    step()
    sync()
    assert np.abs(st["W"].grad.as_numpy() - full_state["W"].grad.as_numpy()).max() / np.abs(full_state["W"].grad.as_numpy()).max() < 1e-5

    # the weight gradient in ROW PANELS: W.grad = X^T @ G is produced 256 rows at a time inside the bucket and every panel
    # is all-reduced on its own (dp.GradSync._produce); the result must equal the full-batch gradient like before
    big_full, big_full_step = workloads.make_cfg4(md, batch=64, d_in=1024, d_out=48, rank=0, world=1)
    big_full_step()
    bst, bstep = workloads.make_cfg4(md, batch=64, d_in=1024, d_out=48, rank=rank, world=world)
    calls = []
    real = comm.allreduce_sum_
    comm.allreduce_sum_ = lambda arr: (calls.append(arr.size), real(arr))[1]
    bsync = dp.GradSync(md, bst["params"], comm, panels=4)
    assert bsync.panels == 4
    for sweep in range(2):
        del calls[:]
        bstep()
        bsync()
        assert calls == [256 * 48] * 3 + [256 * 48 + 48], calls   # four W panels from inside the GEMM loop; the bias (final before W in backward order, next to W in the bucket) rides on the last one
        assert bsync.panel_collectives == 4 * (sweep + 1) and bsync.overlapped == sweep + 1
        for name in ("W", "b"):
            got, exp = bst[name].grad.as_numpy(), big_full[name].grad.as_numpy()
            assert got.shape == exp.shape
            err = np.abs(got - exp).max() / np.abs(exp).max()
            assert err < 1e-5, (name, err)
        # the gradients ARE views of the bucket (no copy of W.grad into it)
        assert np.shares_memory(bst["W"].grad._data, bsync.bucket) and np.shares_memory(bst["b"].grad._data, bsync.bucket)
    bsync.close()
    # panel boundaries that do not divide the weight: K = 1280 in 3 requested panels -> whole 256-row GEMM tiles per panel
    # (512, 512, 256 rows), the short last panel carrying the bias
    ofull, ofull_step = workloads.make_cfg4(md, batch=64, d_in=1280, d_out=24, rank=0, world=1)
    ofull_step()
    ost, ostep = workloads.make_cfg4(md, batch=64, d_in=1280, d_out=24, rank=rank, world=world)
    osync = dp.GradSync(md, ost["params"], comm, panels=3)
    del calls[:]
    ostep()
    osync()
    assert calls == [512 * 24, 512 * 24, 256 * 24 + 24], calls
    for name in ("W", "b"):
        got, exp = ost[name].grad.as_numpy(), ofull[name].grad.as_numpy()
        assert np.abs(got - exp).max() / np.abs(exp).max() < 1e-5, name
        assert np.shares_memory(ost[name].grad._data, osync.bucket)
    osync.close()
    # un-chunked reference of the same sweep: bit-equal on this rank's shard sums? (different GEMM blocking may round
    # differently on a device; on NumPy the panel products are the same dot products) -> equal after the same all-reduce
    usync = dp.GradSync(md, bst["params"], comm, panels=1)
    del calls[:]
    bstep()
    usync()
    assert calls == [1024 * 48 + 48], calls
    assert np.abs(bst["W"].grad.as_numpy() - big_full["W"].grad.as_numpy()).max() / np.abs(big_full["W"].grad.as_numpy()).max() < 1e-5
    usync.close()
    comm.allreduce_sum_ = real
    # a batch that does not split evenly is refused, not truncated
    try:
        workloads.make_cfg4(md, batch=63, d_in=24, d_out=40, rank=rank, world=world)
        raise AssertionError("uneven batch accepted")
    except ValueError:
        pass

    # cfg2: every rank has its own batch block A_r; B.grad = sum_r A_r^T @ G_r, one un-bucketed all-reduce
    st2, step2 = workloads.make_cfg2(md, n=48, rank=rank)
    sync2 = dp.GradSync(md, st2["params"][:1], comm, overlap=False)
    step2()
    local = st2["B"].grad.as_numpy().copy()
    sync2()
    assert sync2.overlapped == 0
    gathered = [torch.zeros(48, 48) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(local))
    exp = sum(g.numpy().astype(np.float64) for g in gathered)
    err = np.abs(st2["B"].grad.as_numpy() - exp).max() / np.abs(exp).max()
    assert err < 1e-6, err
    # A.grad stays local (rows are independent)
    assert st2["A"].grad.shape == (48, 48)
    a_local = st2["A"].grad.as_numpy().copy()
    # overlapped form: B.grad's vjp is served first and its all-reduce is issued before A.grad's vjp runs
    order = []
    real = comm.allreduce_sum_
    comm.allreduce_sum_ = lambda arr: (order.append("allreduce"), real(arr))[1]
    sync3 = dp.GradSync(md, st2["params"][:1], comm)
    step2()
    assert sync3.overlapped == 1 and order == ["allreduce"]
    sync3()
    assert order == ["allreduce"]  # joined, not repeated
    err = np.abs(st2["B"].grad.as_numpy() - exp).max() / np.abs(exp).max()
    assert err < 1e-6, err
    assert np.array_equal(st2["A"].grad.as_numpy(), a_local)
    sync3.close()
    comm.allreduce_sum_ = real
    # RcclComm's rendezvous: rank 0 creates the ncclUniqueId, every rank must call ncclCommInitRank with THAT id,
    # its own rank and the world size (a recording stand-in for libmdhip: no collective runs here)
    import ctypes as C
    from minidiff_amd import _capi

    class _Recorder:
        def __init__(self):
            self.calls = []

        def comm_get_unique_id(self, uid):
            for i in range(_capi.UID_BYTES):
                uid[i] = (37 * i + 11) % 251
            self.calls.append("get_unique_id")

        def comm_init(self, nranks, r, uid):
            self.calls.append(("init", nranks, r, bytes(uid)))

        def comm_destroy(self):
            self.calls.append("destroy")

    rec = _Recorder()
    real_load = _capi.load
    _capi.load = lambda: rec
    try:
        rc = dp.RcclComm(rank, world, dist)
        rc.close()
    finally:
        _capi.load = real_load
    # a failure on ONE rank (rank 0 cannot produce the id; or one rank cannot load librccl) must surface on EVERY rank,
    # before anybody enters ncclCommInitRank, so that all of them can fall back together
    class _FailingId(_Recorder):
        def comm_get_unique_id(self, uid):
            raise RuntimeError("librccl missing (injected)")

    class _FailingProbe(_Recorder):
        def comm_probe(self):
            if rank == 1:
                raise RuntimeError("no RCCL here (injected)")

    for stand_in in (_FailingId, _FailingProbe):
        bad = stand_in()
        _capi.load = lambda bad=bad: bad
        try:
            try:
                dp.RcclComm(rank, world, dist)
                raise AssertionError("rendezvous should have failed on every rank")
            except RuntimeError as e:
                assert "injected" in str(e), e
        finally:
            _capi.load = real_load
        assert not [c for c in bad.calls if isinstance(c, tuple)], bad.calls   # nobody reached comm_init
    flags = [None] * world
    dist.all_gather_object(flags, True)   # the control plane is still in step after the failures
    assert all(flags)
    expected_uid = bytes((37 * i + 11) % 251 for i in range(_capi.UID_BYTES))
    inits = [c for c in rec.calls if isinstance(c, tuple)]
    assert inits == [("init", world, rank, expected_uid)], inits
    assert ("get_unique_id" in rec.calls) == (rank == 0)
    assert rec.calls[-1] == "destroy"

    dist.barrier()
    dist.destroy_process_group()
    print(f"DP-OK rank {rank}/{world}")


if __name__ == "__main__":
    main()
