"""Runs in a subprocess (minidiff parses sys.argv at import): load the UNMODIFIED
reference with `--backend minidiff_amd.plugin`, bound to the CPU test double, and
replay the golden cases through the reference's own Tensor/OpNode tape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MINIDIFF_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

from minidiff_amd import _capi  # noqa: E402

_capi.use_library(os.path.join(ROOT, "oracle", "_build", "libmdhip_host.so"))
sys.argv = [sys.argv[0], "--backend", "minidiff_amd.plugin"]

import numpy as np  # noqa: E402
import minidiff as md  # noqa: E402  (the real reference)
import minidiff.backend as mdb  # noqa: E402
from minidiff_amd import plugin  # noqa: E402
from minidiff_amd.ndarray import DeviceArray  # noqa: E402

plugin.assert_selected()
assert md.__file__.startswith(REF), md.__file__
assert mdb.tensor_class is DeviceArray
assert isinstance(md.Tensor([1.0, 2.0])._data, DeviceArray)

import golden_util as gu  # noqa: E402

# README (README.md:13-36)
x = md.Tensor([[0, 2, -2, 1], [-1, -1, -2, -2]], allow_grad=True)
y = md.Tensor([[2, 3, 4, 5], [0, -1, -3, 2]], allow_grad=True)
f = 2 * y * md.sin(x) - x ** 2
f.backward(allow_higher_order=True)
g = gu.golden()["cfg"]
assert np.allclose(x.grad.as_numpy(), g["cfg1/dx"], rtol=1e-13) and np.allclose(y.grad.as_numpy(), g["cfg1/dy"], rtol=1e-13)
x.grad.backward()
assert np.allclose(x.grad.as_numpy(), g["cfg1/d2x"], rtol=1e-13) and np.allclose(y.grad.as_numpy(), g["cfg1/d2xy"], rtol=1e-13)

n = 0
with np.errstate(all="ignore"):
    for key in gu.case_keys():
        gu.run_case_on(md, key, exact=False)
        n += 1
print(f"DROPIN-OK {n} golden cases through the real reference tape on the plug-in backend")
