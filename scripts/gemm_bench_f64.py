#!/usr/bin/env python3
"""float64 matmul on the f64 matrix cores: direct-to-LDS kernel (k_gemm_f64_glds) against the register-staged one (MDHIP_GEMM_GLDS=0,
read per launch), NN / NT / TN, HIP-event timing, exactness on integer-valued operands first.   usage: gemm_bench_f64.py [MxKxN ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402


def main():
    lib = _capi.load()
    shapes = [(4096, 4096, 4096), (2048, 2048, 2048), (8192, 4096, 4096)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    rng = np.random.default_rng(0)
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    for (M, K, N) in shapes:
        Ai = rng.integers(-3, 4, (M, K)).astype(np.float64)
        Bi = rng.integers(-3, 4, (K, N)).astype(np.float64)
        ref = Ai[:256] @ Bi
        A, B = nd.asarray(rng.standard_normal((M, K))), nd.asarray(rng.standard_normal((K, N)))
        At, Bt = nd.asarray(np.ascontiguousarray(A.get().T)), nd.asarray(np.ascontiguousarray(B.get().T))
        dAi, dBi = nd.asarray(Ai), nd.asarray(Bi)
        dAit, dBit = nd.asarray(np.ascontiguousarray(Ai.T)), nd.asarray(np.ascontiguousarray(Bi.T))
        print(f"M={M} K={K} N={N} float64")
        combos = (("NN", A, B, dAi, dBi), ("NT", A, Bt.T, dAi, dBit.T), ("TN", At.T, B, dAit.T, dBi))
        res = {}
        for glds in ("1", "0"):
            lib.debug_set_option(b"gemm_glds", int(glds))
            for tag, a, b, ai, bi in combos:
                assert np.array_equal(nd.matmul(ai, bi).get()[:256], ref), (tag, glds)
        for _ in range(8):          # pre-roll: clocks and allocator
            nd.matmul(A, B)
        for rnd in range(6):        # the two builds interleaved, round by round: drift shows in both rows
            for glds in ("1", "0"):
                lib.debug_set_option(b"gemm_glds", int(glds))
                for tag, a, b, ai, bi in combos:
                    nd.matmul(a, b)
                    lib.event_record(e0)
                    for _ in range(3):
                        nd.matmul(a, b)
                    lib.event_record(e1)
                    lib.event_elapsed_ms(e0, e1, C.byref(ms))
                    res.setdefault((glds, tag), []).append(3 * 2.0 * M * N * K / (ms.value * 1e-3) / 1e12)
        for glds in ("1", "0"):
            line = ["%s med %5.1f max %5.1f TF" % (t, sorted(res[(glds, t)])[3], max(res[(glds, t)])) for t in ("NN", "NT", "TN")]
            print("   %-44s %s" % ("GLDS=1 (TN: direct-to-LDS; NN / NT: same kernel)" if glds == "1" else "GLDS=0 (register-staged everywhere)", "  ".join(line)))
    lib.debug_set_option(b"gemm_glds", 1)


if __name__ == "__main__":
    main()
