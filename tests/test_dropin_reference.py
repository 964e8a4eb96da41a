"""Drop-in proof (CPU container only): the unmodified reference selects
minidiff_amd.plugin through its own --backend flag and reproduces its own golden
results through its own tape. Skipped where /root/reference does not exist
(the GPU box)."""
import os
import subprocess
import sys

import pytest

REF = os.environ.get("MINIDIFF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "minidiff")), reason="reference checkout not present")
def test_reference_runs_on_plugin_backend(lib, on_gpu):
    if on_gpu:
        pytest.skip("runs against the CPU test double")
    p = subprocess.run([sys.executable, os.path.join(HERE, "dropin_reference_script.py")], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DROPIN-OK" in p.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "minidiff")), reason="reference checkout not present")
def test_reference_fuzz_numpy_backend_vs_plugin(lib, on_gpu, tmp_path):
    """300 random forward+backward programs (tests/fuzz_tape.py) through the real reference's own tape:
    its NumPy backend and the plug-in must agree on every output, gradient and raised exception type."""
    if on_gpu:
        pytest.skip("runs against the CPU test double")
    import pickle

    import numpy as np
    sys.path.insert(0, HERE)
    import fuzz_tape
    outs = {}
    for which, env in (("numpy", {}), ("plugin", {"MDHIP_LAZY": "0"}), ("plugin-lazy", {"MDHIP_LAZY": "1"})):
        path = tmp_path / f"{which}.pkl"
        p = subprocess.run([sys.executable, os.path.join(HERE, "dropin_fuzz_script.py"), which.split("-")[0], "300", "7", str(path)],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        outs[which] = pickle.loads(path.read_bytes())
    assert len(outs["numpy"]) == len(outs["plugin"]) == len(outs["plugin-lazy"]) == 300
    for mode in ("plugin", "plugin-lazy"):   # the fused-chain layer sits under the same unmodified tape
        _compare(outs["numpy"], outs[mode], fuzz_tape, np)


def _compare(ref, got, fuzz_tape, np):
    n_ok = 0
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a[0] == b[0], (i, a[:2], b[:2])
        if a[0] == "raise":
            assert a[1] == b[1], (i, a, b)
            continue
        n_ok += 1
        dt = np.dtype(a[5]).type
        fuzz_tape.close(b[1], a[1], dt, f"case {i} output")
        for k, (ga, gb) in enumerate(zip(a[3], b[3])):
            fuzz_tape.close(gb, ga, dt, f"case {i} grad[{k}]")
        if a[4] is not None:
            for k, (ga, gb) in enumerate(zip(a[4], b[4])):
                fuzz_tape.close(gb, ga, dt, f"case {i} second-order grad[{k}]")
    assert n_ok > 150
