"""Runs in a subprocess (minidiff parses sys.argv at import). Runs the random forward+backward
programs of tests/fuzz_tape.py through the UNMODIFIED reference — `numpy`: on its own NumPy backend;
`plugin`: with `--backend minidiff_amd.plugin` bound to the CPU test double — and pickles outputs,
gradients (second order every fourth case) or the exception type of each case."""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MINIDIFF_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

which, n_cases, seed, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
if which == "plugin":
    from minidiff_amd import _capi
    _capi.use_library(os.path.join(ROOT, "oracle", "_build", "libmdhip_host.so"))
    sys.argv = [sys.argv[0], "--backend", "minidiff_amd.plugin"]
else:
    sys.argv = [sys.argv[0], "--backend", "minidiff.backend.numpy"]

import numpy as np  # noqa: E402
import minidiff as md  # noqa: E402  (the real reference)

assert md.__file__.startswith(REF), md.__file__
if which == "plugin":
    from minidiff_amd import plugin
    plugin.assert_selected()

# fuzz_tape builds engines lazily; only its program generator is used here
import fuzz_tape  # noqa: E402

results = []
for i in range(n_cases):
    second = (i % 4 == 0)
    try:
        with np.errstate(all="ignore"):
            o, l, g1, g2, dt = fuzz_tape.run(md, seed, i, second)
        results.append(("ok", o, l, g1, g2, np.dtype(dt).name))
    except Exception as e:  # the two backends must fail alike
        results.append(("raise", type(e).__name__))
with open(out_path, "wb") as f:
    pickle.dump(results, f)
print(f"DROPIN-FUZZ {which} {len(results)} cases, {sum(r[0] == 'raise' for r in results)} raised")
