"""N>1 path on CPU: two processes, gloo, the sharded sweep + gradient all-reduce
(SURVEY.md §8e). The RCCL communicator itself is covered on the GPU box with a
single-rank communicator (test_rccl_single_rank_gpu)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dp_gloo(world):
    """The sharded sweep + gradient all-reduce at 2, 4 and 8 ranks (the sizes the driver's scaling run uses): shard
    coverage, panel boundaries (even and uneven), bucket views, gradients = full batch, rendezvous failure handling."""
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"DP-OK rank {rank}/{world}" in out


def test_shard_rows():
    from minidiff_amd import dp
    assert dp.shard_rows(8192, 3, 8) == slice(3072, 4096)
    with pytest.raises(ValueError):
        dp.shard_rows(10, 0, 4)


@pytest.mark.gpu
def test_rccl_single_rank_gpu(lib, on_gpu):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce through the C-ABI on the
    library's stream, world size 1 (the only size a one-GPU box allows)."""
    assert on_gpu
    from minidiff_amd import dp, ndarray as nd
    comm = dp.RcclComm(0, 1)
    x = np.random.default_rng(0).standard_normal(1 << 20).astype(np.float32)
    d = nd.asarray(x)
    comm.allreduce_sum_(d)
    assert np.array_equal(d.get(), x)
    di = nd.asarray(np.arange(1000))
    comm.allreduce_sum_(di)
    assert np.array_equal(di.get(), np.arange(1000))
    comm.close()


@pytest.mark.gpu
def test_overlapped_allreduce_single_rank_gpu(engines, on_gpu):
    """The collective issued from inside backward() on the second stream (ready/done events), world
    size 1: gradients equal the un-overlapped sweep bit for bit and the streams are joined."""
    assert on_gpu
    from minidiff_amd import dp, workloads
    hip, _ = engines
    comm = dp.RcclComm(0, 1)
    try:
        for maker, kw, names in ((workloads.make_cfg2, {"n": 384}, ("A", "B")),
                                 (workloads.make_cfg4, {"batch": 256, "d_in": 192, "d_out": 160}, ("W", "b"))):
            st, step = maker(hip, **kw)
            step()
            ref = {k: st[k].grad.as_numpy().copy() for k in names}
            params = st["params"][:1] if maker is workloads.make_cfg2 else st["params"]
            sync = dp.GradSync(hip, params, comm, force=True)
            for _ in range(3):
                step()
                sync()
            assert sync.overlapped == 3
            for k in names:
                np.testing.assert_array_equal(st[k].grad.as_numpy(), ref[k])
            sync.close()
            step()
            assert sync.overlapped == 3  # hooks removed
    finally:
        comm.close()


@pytest.mark.gpu
def test_segmented_graph_replay_single_rank_gpu(engines, on_gpu):
    """graph.SegmentedSweep: the data-parallel sweep replayed as hipGraph segments with the collectives between them
    (world size 1, RCCL): same gradients as the eager sweep bit for bit, new inputs fed INTO the resident arrays are
    picked up by a replay, and no Python tape runs during a replay."""
    assert on_gpu
    from minidiff_amd import dp, workloads
    from minidiff_amd.graph import SegmentedSweep
    hip, _ = engines
    comm = dp.RcclComm(0, 1)
    try:
        st, step = workloads.make_cfg4(hip, batch=512, d_in=1024, d_out=256)
        sync = dp.GradSync(hip, st["params"], comm, force=True, panels=2)

        def sweep():
            out = step()
            sync()
            return out

        sweep()
        ref = {k: st[k].grad.as_numpy().copy() for k in ("W", "b")}
        seg = SegmentedSweep(sweep, comm)
        # two panels (the second carries the bias) + the join: 3 communicator calls, a graph segment in front of each and one behind
        assert seg.calls == 3 and seg.segments == 4, (seg.calls, seg.segments)
        overlapped = sync.overlapped
        for _ in range(3):
            out = seg.replay()
        assert sync.overlapped == overlapped            # replays run no Python
        for k in ("W", "b"):
            np.testing.assert_array_equal(out[k].grad.as_numpy(), ref[k])
        # new batch written into the resident input: the replay computes ITS gradients
        x2 = np.random.default_rng(11).standard_normal((512, 1024)).astype(np.float32)
        st["X"]._data[...] = x2
        out = seg.replay()
        got = {k: out[k].grad.as_numpy().copy() for k in ("W", "b")}
        seg.close()
        sweep()
        for k in ("W", "b"):
            np.testing.assert_array_equal(got[k], st[k].grad.as_numpy())
        assert not np.array_equal(got["W"], ref["W"])
        sync.close()
    finally:
        comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("comm", ["rccl", "torch"])
@pytest.mark.parametrize("workload", ["cfg2", "cfg4"])
def test_bench_distributed_path_on_one_gpu(on_gpu, comm, workload):
    """bench.py's N>1 branch end to end at world size 1 (MDHIP_BENCH_FORCE_DIST): process group,
    ncclUniqueId exchange, RCCL communicator (direct, and through torch.distributed), the
    per-sweep all-reduce of the gradient (bucketed for cfg4) and max-over-ranks timing."""
    assert on_gpu
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MDHIP_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0",
               MDHIP_DP_GRAPH="1")    # (segments forced: by default a three-sweep trial picks between them and eager sweeps)
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="mdhip_bench_"), "detail.json")
    def launch():
        return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                               "--workload", workload, "--size", "512", "--no-cpu-baseline", "--comm", comm, "--detail", detail]
                              + (["--allow-torch-comm"] if comm == "torch" else []),
                              env=env, capture_output=True, text=True, timeout=600)
    p = launch()
    if p.returncode != 0:
        # This test failed ONCE (the first GPU job of a cold box, round 4) and passed on every run before and after; its output was
        # lost. Keep the whole story where a later reader finds it, and give the rendezvous one more go on a fresh port — a child
        # that EXITED with an error, not a hang (the run above is bounded by its own timeout).
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", f"bench_dist_failure_{workload}_{comm}.log"), "w") as f:
            f.write(f"rc {p.returncode}\n--- stdout\n{p.stdout}\n--- stderr\n{p.stderr}\n")
        print(f"[test_dp_gloo] first attempt failed (rc {p.returncode}); log kept in gpurun_out/; stderr tail:\n{p.stderr[-1500:]}", file=sys.stderr)
        env["MASTER_PORT"] = str(_free_port())
        p = launch()
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    head = json.load(open(detail))["head"]
    assert line["config"]["collective"] == ("rccl-direct" if comm == "rccl" else "rccl-torch"), line["config"]
    assert line["config"]["rccl_ranks"] == (1 if comm == "rccl" else None)      # what ncclCommCount reports for the live communicator
    assert line["config"]["allreduce_bytes"] > 0 and line["value"] > 0
    # every sweep that ran through Python sent its collective(s) from inside backward(): pre-roll, 1 + 1 eager sweeps in front of
    # the capture, the capturing run itself (the K timed sweeps are graph-segment replays: no Python), ten single synchronised
    # sweeps and the detail pass of cfg4. The torch communicator has no asynchronous form, but is cut into segments all the same.
    seg = head["config"]["graph_replay"]
    assert line["config"]["graph_replay"] is True and line["config"]["graph_segments"] == seg["segments"]
    # (cfg4 over RCCL: the weight-gradient panels + the join; the torch communicator has one synchronous all-reduce and no join)
    assert isinstance(seg, dict) and seg["segments"] >= 2 and seg["collective_calls"] >= (2 if comm == "rccl" else 1), seg
    assert head["config"]["allreduce_overlapped_sweeps"] == head["preroll_sweeps"] + 1 + 1 + 1 + 10 + (3 if workload == "cfg4" else 0)


@pytest.mark.gpu
def test_bench_distributed_path_picks_its_sweep_mode(on_gpu):
    """Without MDHIP_DP_GRAPH the N > 1 branch times three eager sweeps against three segmented replays before the timed region and
    runs the faster; the line says which (`graph_replay`) and carries both trial times."""
    assert on_gpu
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MDHIP_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MDHIP_DP_GRAPH", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--size", "1024", "--no-cpu-baseline", "--no-secondary"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:      # (as in test_bench_distributed_path_on_one_gpu: log, one more attempt on a fresh port)
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "bench_dist_failure_trial.log"), "w") as f:
            f.write(f"rc {p.returncode}\n--- stdout\n{p.stdout}\n--- stderr\n{p.stderr}\n")
        env["MASTER_PORT"] = str(_free_port())
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
    cfg = json.loads(p.stdout.strip().splitlines()[-1])["config"]
    assert cfg["workload"].startswith("cfg2") and cfg["collective"] == "rccl-direct"
    assert cfg["sweep_trial_eager_ms"] > 0 and cfg["sweep_trial_segments_ms"] > 0
    assert cfg["graph_replay"] is (cfg["sweep_trial_segments_ms"] < cfg["sweep_trial_eager_ms"])
