#!/usr/bin/env python3
"""A/B the fp32 MFMA GEMM tile configs in ONE process (guide §5.4 rule 24):
interleaved rounds over configs x layouts x shapes, HIP-event timing, random data.
Tiles / staging / buffers are forced through the option table (mdhip_debug_set_option, csrc/md_options.h)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

CFGS = {-1: "auto (tile picker)", 0: "128x128x16", 1: "64x64x16", 2: "128x64x16", 3: "256x128x16", 4: "256x256x32 (TN)", 5: "128x128x32 (TN)", 6: "128x64x32 (TN)", 7: "128x128 8 waves"}
if os.environ.get("GEMM_CFGS"):
    CFGS = {int(k): CFGS[int(k)] for k in os.environ["GEMM_CFGS"].split(",")}
if os.environ.get("GEMM_NBUF_AB"):   # every config twice: two / three LDS buffers in the k-contiguous direct-to-LDS kernel (option gemm_nbuf)
    CFGS = {(99 if k == -1 else k) + 1000 * g: v + (" %dbuf" % (g + 2)) for k, v in CFGS.items() for g in (0, 1)}
if os.environ.get("GEMM_PEEL_AB"):   # every config twice: single ragged launch / peeled (option gemm_peel)
    CFGS = {(99 if k == -1 else k) + 10000 * g: v + (" peel" if g else " one launch") for k, v in CFGS.items() for g in (0, 1)}
if os.environ.get("GEMM_GLDS_AB"):   # every config twice: register staging / direct-to-LDS staging (option gemm_glds)
    CFGS = {(99 if k == -1 else k) + 100 * g: v + (" +glds" if g else "") for k, v in CFGS.items() for g in (0, 1)}   # (99 = auto)


REPS = int(os.environ.get("GEMM_REPS", "5"))   # launches per timing (small shapes: more, the gaps between 5 short launches weigh)


def main():
    lib = _capi.load()
    opt = lambda name, v: lib.debug_set_option(name.encode(), int(v))   # noqa: E731
    shapes = [(4096, 4096, 4096), (2048, 2048, 2048), (8192, 4096, 4096), (1024, 4096, 4096), (4096, 1024, 4096)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    rng = np.random.default_rng(0)
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    ms = C.c_float()
    for (M, K, N) in shapes:
        A = nd.asarray(rng.standard_normal((M, K), dtype=np.float32))
        B = nd.asarray(rng.standard_normal((K, N), dtype=np.float32))
        At = nd.asarray(np.ascontiguousarray(A.get().T))
        Bt = nd.asarray(np.ascontiguousarray(B.get().T))
        ref = None
        res = {}
        rounds = int(os.environ.get("GEMM_ROUNDS", "5"))
        for _ in range(6):          # pre-roll: the first milliseconds after an idle gap run at ramping clocks (the med 126 / max 150
            nd.matmul(A, B)         # spread of round 2's log was the first round of each shape, taken cold)
        for rnd in range(rounds):
            for cfg in CFGS:
                if os.environ.get("GEMM_NBUF_AB"):
                    opt("gemm_nbuf", 2 + (cfg // 1000) % 10)
                if os.environ.get("GEMM_PEEL_AB"):
                    opt("gemm_peel", (cfg // 10000) % 10)
                if cfg % 100 == 99 or cfg == -1:
                    opt("gemm_cfg", -1)      # the library's own choice
                else:
                    opt("gemm_cfg", cfg % 100)
                if os.environ.get("GEMM_GLDS_AB"):
                    opt("gemm_glds", (cfg // 100) % 10)
                combos = (("NN", A, B), ("NT", A, Bt.T), ("TN", At.T, B)) + ((("TT", At.T, Bt.T),) if os.environ.get("GEMM_TT") else ())
                for tag, a, b in combos:
                    nd.matmul(a, b)  # warm
                    nd.matmul(a, b)
                    lib.event_record(e0)
                    for _ in range(REPS):
                        out = nd.matmul(a, b)
                    lib.event_record(e1)
                    lib.event_elapsed_ms(e0, e1, C.byref(ms))
                    tf = REPS * 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
                    res.setdefault((cfg, tag), []).append(tf)
                    if rnd == 0:
                        h = out.get()
                        if ref is None:
                            ref = h
                        assert np.abs(h - ref).max() / np.abs(ref).max() < 5e-6, (cfg, tag)
        print(f"M={M} K={K} N={N}")
        for cfg, name in CFGS.items():
            print("   %-22s " % name + "  ".join("%s med %6.1f min %6.1f max %6.1f TF" % (t, sorted(res[(cfg, t)])[len(res[(cfg, t)]) // 2], min(res[(cfg, t)]), max(res[(cfg, t)])) for t in (("NN", "NT", "TN", "TT") if os.environ.get("GEMM_TT") else ("NN", "NT", "TN"))))
    opt("gemm_cfg", -1); opt("gemm_glds", 1); opt("gemm_nbuf", 0); opt("gemm_peel", 1)


if __name__ == "__main__":
    main()
