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
