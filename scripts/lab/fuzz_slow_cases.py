"""Which cases of tests/fuzz_device.py are slow on the device? usage: fuzz_slow_cases.py SEED FIRST LAST [--narrow] [--big] (lab script)"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_device
seed, first, last = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if "--narrow" in sys.argv:
    fuzz_device.DTYPES.extend(fuzz_device.NARROW_DTYPES)
big = "--big" in sys.argv
from minidiff_amd import ndarray as nd
for i in range(first, last):
    rng = np.random.default_rng([seed, i])
    t = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    try:
        fuzz_device.one_case(rng, big)
    except Exception as e:
        print("case", i, "raised", type(e).__name__, str(e)[:100], flush=True)
    pr.disable()
    nd._lib().sync()
    dt = time.perf_counter() - t
    if dt > 1.0:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
        print(f"case {i}: {dt:.1f} s\n" + "\n".join(l for l in s.getvalue().splitlines() if "ndarray.py" in l or "fuzz_device" in l or "_capi" in l)[:3000], flush=True)
    if (i + 1) % 100 == 0:
        print("..", i + 1, flush=True)
