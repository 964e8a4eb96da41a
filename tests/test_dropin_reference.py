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
