"""bench.py's output contract (one JSON line, agreed keys and types), exercised on the CPU test double at toy
sizes so that a broken bench is caught here and not on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


# (cfg4 keeps its 4096 x 4096 weight at any batch: too much for the double's triple-loop GEMM)
@pytest.mark.parametrize("workload,extra", [("cfg2", []), ("cfg3", []), ("cfg5", []), ("cfg3", ["--lazy"])])
def test_bench_prints_one_contract_line(lib, on_gpu, workload, extra):
    if on_gpu:
        pytest.skip("CPU-double check; the GPU box runs the real bench")
    size = {"cfg2": "64", "cfg3": "4096", "cfg5": "64"}[workload]
    p = subprocess.run([sys.executable, os.path.join(HERE, "bench_contract_script.py"), "--workload", workload, "--size", size,
                        "--steps", "2", "--warmup", "1"] + extra, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["metric"].startswith("forward+backward passes/sec") and d["unit"] == "passes/s" and d["dtype"] == "f32"
    assert d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
    r = d["roofline"]
    assert r is None or {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["value"] > 0
    # the metric's second half: gradients of the headline sweep against the NumPy engine, norm-wise (<= 1e-5 for fp32)
    g = d["grad_linf_rel"]
    assert set(g) == {"cfg2": {"A", "B"}, "cfg3": {"x", "y"}, "cfg5": {"A", "B"}}[workload]
    assert all(0.0 <= v <= 1e-5 for v in g.values()), g


def test_bench_two_ranks_control_flow_over_gloo(lib, on_gpu):
    """bench.py's N > 1 branch with two real processes (gloo control plane, HostComm on the CPU double's memory): cfg4 is the
    default workload, strong scaling, tensors_per_s, the weight gradient all-reduced in row panels, then the weak-scaling cfg2
    under `secondary` with a second communicator. Numbers mean nothing; the flow and the line's shape do."""
    if on_gpu:
        pytest.skip("CPU-double check")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MDHIP_BENCH_HOST_COMM="1", MDHIP_BENCH_CFG4_DIM="512", MDHIP_DP_PANELS="2", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "bench_contract_script.py"), "--gpus", "2", "--size", "64",
                                       "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for rank, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}:\n{out[-1500:]}\n{err[-3000:]}"
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]   # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["workload"].startswith("cfg4") and d["tensors_per_s"] == pytest.approx(64 * d["value"])
    assert d["config"]["allreduce_panels"] == 2 and d["config"]["allreduce_bytes"] == (512 * 512 + 512) * 4
    # every sweep (pre-roll + warm-up + timed + the per-kernel detail pass) sent its collectives from inside backward()
    assert d["config"]["allreduce_overlapped_sweeps"] == d["preroll_sweeps"] + 1 + 2 + 10 + 2   # (+ ten single synchronised sweeps)
    sec = d["secondary"]["cfg2_weak"]
    assert "error" not in sec and sec["scaling"] == "weak" and sec["config"]["collective"] == "gloo-host(test)" and sec["value"] > 0
    # the same job's N = 1 figure of the headline workload (un-sharded batch, no collective)
    solo = d["single_gpu_same_workload"]
    assert solo["n_gpus"] == 1 and solo["value"] > 0 and solo["workload"].startswith("cfg4") and solo["tensors_per_s"] == pytest.approx(64 * solo["value"])
