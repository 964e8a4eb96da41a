#!/usr/bin/env python3
"""GB/s of elementwise / broadcast / reduction calls at shapes OTHER than the BASELINE configs
(looking for performance cliffs: strided views, middle-axis reductions, column broadcasts, f64,
ints). HIP events, algorithmic bytes (each input once, output once)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402


def main():
    lib = _capi.load()
    rng = np.random.default_rng(0)
    R, Cc = 8192, 4096
    z = nd.asarray(rng.standard_normal((R, Cc), dtype=np.float32))
    w = nd.asarray(rng.standard_normal((R, Cc), dtype=np.float32))
    zt = nd.asarray(rng.standard_normal((Cc, R), dtype=np.float32))
    col = nd.asarray(rng.standard_normal((R, 1), dtype=np.float32))
    t3 = nd.asarray(rng.standard_normal((64, 512, 1024), dtype=np.float32))
    t4 = nd.asarray(rng.standard_normal((32, 16, 256, 256), dtype=np.float32))
    zd = nd.asarray(rng.standard_normal((R // 2, Cc)))
    zi = nd.asarray(rng.integers(-100, 100, (R // 2, Cc)))
    idx = nd.asarray(rng.integers(0, R, (R,)))
    perm = nd.asarray(rng.permutation(R))
    idx64 = nd.asarray(rng.integers(0, 64, (R,)))
    E = 4 * R * Cc
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    ms = C.c_float()
    cases = [
        ("add(z, w) contiguous", lambda: nd.add(z, w), 3 * E),
        ("add(z, col (R,1))", lambda: nd.add(z, col), 2 * E),
        ("add(z, zt.T) transposed view", lambda: nd.add(z, zt.T), 3 * E),
        ("copy(zt.T) transpose", lambda: nd.copy(zt.T), 2 * E),
        ("mul(z[:, ::2], w[:, ::2])", lambda: nd.multiply(z[:, ::2], w[:, ::2]), 3 * E // 2),
        ("mul(z[::2], w[::2]) row-strided", lambda: nd.multiply(z[::2], w[::2]), 3 * E // 2),
        ("exp f64", lambda: nd.exp(zd), 2 * 8 * zd.size),
        ("add int64", lambda: nd.add(zi, zi), 3 * 8 * zi.size),
        ("astype f32->f64", lambda: nd.astype(z, np.float64), 12 * z.size),
        ("sum axis=1 (rows)", lambda: nd.sum(z, axis=1), E),
        ("sum axis=0 (cols)", lambda: nd.sum(z, axis=0), E),
        ("max axis=1", lambda: nd.max(z, axis=1), E),
        ("max axis=0", lambda: nd.max(z, axis=0), E),
        ("argmax axis=1", lambda: nd.argmax(z, axis=1), E),
        ("argmax axis=0", lambda: nd.argmax(z, axis=0), E),
        ("mean axis=0", lambda: nd.mean(z, axis=0), E),
        ("std axis=1", lambda: nd.std(z, axis=1), E),
        ("std axis=0", lambda: nd.std(z, axis=0), E),
        ("sum 3d axis=1 (middle)", lambda: nd.sum(t3, axis=1), 4 * t3.size),
        ("sum 3d axis=(0,2)", lambda: nd.sum(t3, axis=(0, 2)), 4 * t3.size),
        ("sum 3d axis=2", lambda: nd.sum(t3, axis=2), 4 * t3.size),
        ("sum 4d axis=(0,2,3) (bn-style)", lambda: nd.sum(t4, axis=(0, 2, 3)), 4 * t4.size),
        ("sum 4d axis=1", lambda: nd.sum(t4, axis=1), 4 * t4.size),
        ("sum(zt.T) full, transposed view", lambda: nd.sum(zt.T), E),
        ("sum f64 all", lambda: nd.sum(zd), 8 * zd.size),
        ("gather rows z[idx]", lambda: z[idx], 2 * E),
        ("index_add rows (unique idx)", lambda: nd.index_add(w, perm, z), 3 * E),
        ("index_add rows (dups, 64 dests)", lambda: nd.index_add(w, idx64, z), 2 * E),
        ("z[perm] = w  (row assignment)", lambda: z.__setitem__(perm, w), 2 * E),
        ("prod axis=1", lambda: nd.prod(z, axis=1), E),
        ("any(z > 3)", lambda: nd.any(nd.greater(z, 3)), E + z.size + z.size),
    ]
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, fn, nbytes in cases:
        if only and only not in name:
            continue
        for _ in range(2):
            fn()
        best = 1e9
        for _ in range(3):
            lib.event_record(e0)
            for _ in range(3):
                fn()
            lib.event_record(e1)
            lib.event_elapsed_ms(e0, e1, C.byref(ms))
            best = min(best, ms.value / 3)
        print("%-36s %8.3f ms  %7.1f GB/s" % (name, best, nbytes / best / 1e6), flush=True)


if __name__ == "__main__":
    main()
