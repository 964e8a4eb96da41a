import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from minidiff_amd import ndarray as nd
rng = np.random.default_rng(1)
def ulp(got, ref):
    ref32 = ref.astype(np.float32)
    return (np.abs(got.astype(np.float64) - ref) / np.spacing(np.abs(ref32)).astype(np.float64)).max()
for name, x in (("normal", rng.standard_normal(8_000_000)), ("[-1e5,1e5]", rng.uniform(-1e5, 1e5, 4_000_000)), ("[-1e6,1e6] (OCML path above 105615)", rng.uniform(-1e6, 1e6, 4_000_000)),
                ("near k*pi/2", (np.arange(1, 60000) * (np.pi / 2)).repeat(20) * (1 + rng.uniform(-3e-7, 3e-7, 59999 * 20))), ("huge 1e9", rng.uniform(-1e9, 1e9, 1_000_000)),
                ("specials", np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 105615.0, 105616.0, -105615.0, 1e-30, -1e-30, 3.4e38]))):
    x = x.astype(np.float32)
    d = nd.asarray(x)
    s, c = nd.sin(d).get(), nd.cos(d).get()
    xs = x.astype(np.float64)
    with np.errstate(all="ignore"):
        rs, rc = np.sin(xs), np.cos(xs)
    fin = np.isfinite(rs)
    assert np.array_equal(np.isnan(s), np.isnan(rs.astype(np.float32))) and np.array_equal(np.isnan(c), np.isnan(rc.astype(np.float32))), name
    if name == "specials":
        assert np.signbit(s[1]) and s[1] == 0 and s[0] == 0 and not np.signbit(s[0]) and c[0] == 1 and c[1] == 1
    print(f"{name:40s} sin max ulp {ulp(s[fin], rs[fin]):.3f}  cos max ulp {ulp(c[fin], rc[fin]):.3f}   vs numpy float32 loops: sin {np.abs(s[fin]-np.sin(x)[fin]).max():.2e} cos {np.abs(c[fin]-np.cos(x)[fin]).max():.2e}")
print("sincos ok")
