"""MDHIP_TRACE=<file>: the opt-in call log at the C-ABI shim."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from minidiff_amd import _capi
if %(double)r:
    _capi.use_library(%(double)r)
from minidiff_amd import ndarray as nd
a = nd.asarray(np.arange(12, dtype=np.float32).reshape(3, 4))
b = nd.multiply(a, a.T[:3, :3][:, :1])        # a strided, broadcast operand
c = nd.sum(nd.sin(b), axis=0)
print(np.asarray(c).shape)
"""


def test_call_log_lists_entry_points_with_shapes(tmp_path, lib, on_gpu):
    from conftest import HOST_DOUBLE
    log = tmp_path / "calls.jsonl"
    env = dict(os.environ, MDHIP_TRACE=str(log), MDHIP_LAZY="0")  # the eager entry points are what the log is checked for
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "double": "" if on_gpu else HOST_DOUBLE}],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    recs = [json.loads(line) for line in log.read_text().splitlines()]
    names = [r["call"] for r in recs]
    for expected in ("init", "alloc", "h2d", "binary", "unary", "reduce", "d2h"):   # (frees go straight to the C symbol)
        assert expected in names, (expected, names)
    mul = next(r for r in recs if r["call"] == "binary")
    arrays = [a for a in mul["args"] if isinstance(a, dict)]
    assert arrays[0]["shape"] == [3, 4] and arrays[0]["strides"] == [4, 1]
    assert arrays[1]["shape"] == [3, 4] and arrays[1]["strides"][1] == 0      # the broadcast column
    assert all(r["us"] >= 0 for r in recs)
