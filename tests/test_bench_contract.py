"""bench.py's output contract (one JSON line under 2 KB, agreed keys and types, flat per-config keys inside `roofline`, the
full detail in a side file), its self-spawning N > 1 launch and its loud failures — exercised on the CPU test double at toy
sizes so that a broken bench is caught here and not on the GPU box."""
import json
import os
import subprocess
import sys
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = os.path.join(HERE, "bench_contract_script.py")


def _no_rank_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _one_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    assert len(lines[0]) <= 2000, len(lines[0])      # the driver keeps the last 2 KB of stdout
    return json.loads(lines[0])


# (cfg4 keeps its 4096 x 4096 weight at any batch: too much for the double's triple-loop GEMM)
@pytest.mark.parametrize("workload,extra", [("cfg2", []), ("cfg3", []), ("cfg5", []), ("cfg3", ["--lazy"])])
def test_bench_prints_one_contract_line(lib, on_gpu, workload, extra, tmp_path):
    if on_gpu:
        pytest.skip("CPU-double check; the GPU box runs the real bench")
    size = {"cfg2": "64", "cfg3": "4096", "cfg5": "64"}[workload]
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, SCRIPT, "--workload", workload, "--size", size, "--steps", "2", "--warmup", "1",
                        "--detail", str(detail)] + extra, capture_output=True, text=True, timeout=600, env=_no_rank_env())
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    d = _one_line(p.stdout)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["metric"].startswith("forward+backward passes/sec") and d["unit"] == "passes/s" and d["dtype"] == "f32"
    assert d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
    assert all(not isinstance(v, (dict, list)) for v in d["config"].values())          # scalars only: what the driver's record keeps
    assert all(len(v) <= 120 for v in d["config"].values() if isinstance(v, str))
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["value"] > 0
    # the metric's second half: gradients of the sweep against the NumPy engine, norm-wise (<= 1e-5 for fp32)
    assert 0.0 <= d["grad_linf_rel_max"] <= 1e-5 and r["grad_linf_rel_max"] == d["grad_linf_rel_max"]
    full = json.loads(detail.read_text())
    name = workload + ("_lazy" if extra else "")
    assert set(full["grad_linf_rel"][name]) == {"cfg2": {"A", "B"}, "cfg3": {"x", "y"}, "cfg5": {"A", "B"}}[workload]
    assert full["line"]["value"] == pytest.approx(d["value"], rel=1e-3) and full["head"]["preroll_sweeps"] > 0


def test_bench_flat_per_config_keys(lib, on_gpu, tmp_path):
    """The other BASELINE configs ride INSIDE `roofline` as flat scalar keys (the driver's record drops nested dicts), the NumPy
    engine is timed for every config, and the line still fits 2 KB."""
    if on_gpu:
        pytest.skip("CPU-double check")
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, SCRIPT, "--size", "64", "--steps", "2", "--warmup", "1", "--force-secondary", "--detail", str(detail)],
                       capture_output=True, text=True, timeout=900, env=_no_rank_env(MDHIP_BENCH_CFG4_DIM="64"))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    d = _one_line(p.stdout)
    r = d["roofline"]
    for key in ("cfg3_passes_per_s", "cfg3_mul_frac", "cfg3_mul_us", "cfg3_mul_bytes", "cfg3_whole_sweep_frac", "cfg3_lazy_passes_per_s",
                "cfg3_lazy_frac", "cfg4_passes_per_s", "cfg4_gemm_frac", "cfg4_pair_frac", "cfg4_pair_us", "cfg4_pair_bytes",
                "cfg4_loss_sum_frac", "cfg4_colsum_frac", "cfg4_maskprod_frac", "cfg4_lazy_passes_per_s", "cfg5_passes_per_s", "cfg5_frac",
                "grad_linf_rel_max", "cpu_cfg2_passes_per_s", "cpu_cfg3_passes_per_s", "cpu_cfg4_passes_per_s", "cpu_cfg5_passes_per_s"):
        assert isinstance(r.get(key), (int, float)), (key, sorted(r))
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    # a fraction is recomputable from the record alone: bytes / us / 8 TB/s
    assert r["cfg4_pair_frac"] == pytest.approx(r["cfg4_pair_bytes"] / (r["cfg4_pair_us"] * 1e-6) / 8e12, rel=2e-3)
    assert r["cfg3_mul_frac"] == pytest.approx(r["cfg3_mul_bytes"] / (r["cfg3_mul_us"] * 1e-6) / 8e12, rel=2e-3)
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cfg3_value"] > 0 and c["cfg4_value"] > 0 and c["cfg5_value"] > 0
    assert d["grad_linf_rel_max"] <= 1e-5
    full = json.loads(detail.read_text())
    assert set(full["secondary"]) == {"cfg3", "cfg3_lazy", "cfg4", "cfg4_lazy", "cfg5"} and not any("error" in v for v in full["secondary"].values())
    assert set(full["grad_linf_rel"]) == {"cfg2", "cfg3", "cfg3_lazy", "cfg4", "cfg4_lazy", "cfg5"}


def _check_two_rank_line(d, full):
    # the headline at N > 1 is the metric's own workload (cfg2), weak scaling over batch rows: value = N x sweeps/s
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["workload"].startswith("cfg2") and d["tensors_per_s"] == pytest.approx(64 * d["value"], rel=1e-3)
    assert d["value"] == pytest.approx(2 * 1e3 / d["ms_per_step"], rel=1e-3)
    assert d["config"]["allreduce_bytes"] == 64 * 64 * 4 and d["config"]["allreduce_panels"] >= 1
    assert d["config"]["collective"] == "gloo-host(test)" and d["config"]["rccl_ranks"] is None
    head = full["head"]
    # every sweep (pre-roll + warm-up + timed + ten single synchronised sweeps) sent its collective from inside backward()
    assert head["config"]["allreduce_overlapped_sweeps"] == head["preroll_sweeps"] + 1 + 2 + 10
    # the same job's N = 1 figure of the headline workload (no collective)
    solo = full["single_gpu_same_workload"]
    assert solo["n_gpus"] == 1 and solo["value"] > 0 and solo["workload"].startswith("cfg2") and solo["tensors_per_s"] == pytest.approx(64 * solo["value"])
    assert d["single_gpu_value"] == pytest.approx(solo["value"], rel=1e-3)
    # then BASELINE's configs[3]: ONE global batch split over the ranks (strong), its bucket all-reduced in row panels
    sec = full["secondary"]["cfg4_strong"]
    assert "error" not in sec and sec["scaling"] == "strong" and sec["config"]["collective"] == "gloo-host(test)" and sec["value"] > 0
    assert sec["config"]["allreduce_panels"] == 2 and sec["config"]["allreduce_bytes"] == (512 * 512 + 512) * 4
    assert sec["tensors_per_s"] == pytest.approx(64 * sec["value"], rel=1e-3)
    r = d["roofline"]
    assert r["cfg4_strong_passes_per_s"] == pytest.approx(sec["value"], rel=1e-3)
    assert r["cfg4_strong_single_gpu_passes_per_s"] == pytest.approx(sec["single_gpu_same_workload"]["value"], rel=1e-3)
    assert r["cfg4_strong_tensors_per_s"] == pytest.approx(sec["tensors_per_s"], rel=1e-3) and r["cfg4_strong_ms_per_step"] > 0


TWO_RANK_ENV = dict(MDHIP_BENCH_HOST_COMM="1", MDHIP_BENCH_CFG4_DIM="512", MDHIP_DP_PANELS="2", OMP_NUM_THREADS="2")
TWO_RANK_ARGS = ["--gpus", "2", "--size", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]


def test_bench_two_ranks_control_flow_over_gloo(lib, on_gpu, tmp_path):
    """bench.py's N > 1 branch started the way torch.distributed.run starts it (rank environment given): two real processes
    (gloo control plane, HostComm on the CPU double's memory): cfg2 — the metric's workload — is the headline at every N (weak scaling
    over batch rows, B.grad all-reduced), then the strong-scaling cfg4 with a second communicator and its own single-GPU figure. Numbers mean nothing;
    the flow and the line's shape do."""
    if on_gpu:
        pytest.skip("CPU-double check")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    detail = tmp_path / "detail.json"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **TWO_RANK_ENV)
        procs.append(subprocess.Popen([sys.executable, SCRIPT] + TWO_RANK_ARGS + ["--detail", str(detail)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for rank, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}:\n{out[-1500:]}\n{err[-3000:]}"
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]   # rank 0 alone prints
    _check_two_rank_line(_one_line(outs[0][0]), json.loads(detail.read_text()))


def test_bench_starts_its_own_ranks(lib, on_gpu, tmp_path):
    """`python bench.py --gpus 2` with NO rank environment — the way the driver ran N = 1 — must start its two
    ranks itself and print rank 0's single line with n_gpus == 2."""
    if on_gpu:
        pytest.skip("CPU-double check")
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, SCRIPT] + TWO_RANK_ARGS + ["--detail", str(detail)], capture_output=True, text=True, timeout=900,
                       env=_no_rank_env(**TWO_RANK_ENV))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    _check_two_rank_line(_one_line(p.stdout), json.loads(detail.read_text()))


def test_bench_fails_loudly_when_a_rank_dies(lib, on_gpu):
    """... and when one of its ranks dies (here: at once, leaving rank 0 waiting in the rendezvous) the parent kills the others and
    returns a non-zero code well inside the limit — no line, no hang."""
    if on_gpu:
        pytest.skip("CPU-double check")
    t0 = time.time()
    p = subprocess.run([sys.executable, SCRIPT] + TWO_RANK_ARGS + ["--rank-timeout", "120"], capture_output=True, text=True, timeout=300,
                       env=_no_rank_env(MDHIP_TEST_DIE_RANK="1", **TWO_RANK_ENV))
    assert p.returncode != 0 and time.time() - t0 < 100, (p.returncode, time.time() - t0)
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "rank 1 exited with code 7" in p.stderr, p.stderr[-2000:]


def test_bench_rank_timeout(lib, on_gpu):
    """A job that exceeds --rank-timeout is killed (every rank) and reported with exit code 124."""
    if on_gpu:
        pytest.skip("CPU-double check")
    p = subprocess.run([sys.executable, SCRIPT] + TWO_RANK_ARGS + ["--rank-timeout", "0.5"], capture_output=True, text=True, timeout=300,
                       env=_no_rank_env(**TWO_RANK_ENV))
    assert p.returncode == 124 and "no result within --rank-timeout" in p.stderr, (p.returncode, p.stderr[-1500:])


def test_bench_refuses_the_torch_fallback_at_n_gt_1(lib, on_gpu):
    """At N > 1 anything but the library's own RCCL communicator is an error unless --allow-torch-comm is given: here the direct
    communicator cannot be built (the CPU double has no collective backend), so every rank must exit non-zero with the reason."""
    if on_gpu:
        pytest.skip("CPU-double check")
    env = {k: v for k, v in TWO_RANK_ENV.items() if k != "MDHIP_BENCH_HOST_COMM"}
    p = subprocess.run([sys.executable, SCRIPT] + TWO_RANK_ARGS + ["--no-secondary", "--rank-timeout", "300"], capture_output=True, text=True, timeout=600,
                       env=_no_rank_env(**env))
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "refusing to fall back" in p.stderr, p.stderr[-2500:]
