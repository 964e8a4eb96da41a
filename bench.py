#!/usr/bin/env python3
"""bench.py — forward+backward passes/sec of the minidiff hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|cfg4|cfg5] [--lazy] [--graph]

Metric (BASELINE.json): forward+backward passes/sec on the 4096x4096 fp32 matmul
graph. At N = 1 the headline workload is configs[1] (cfg2): C = A @ B;
C.backward() — 3 GEMMs (NN, NT, TN), 412,316,860,416 FLOP per sweep, inputs
resident in HBM. After the headline's timed region the same process runs the other
BASELINE configs (cfg3 eager / lazy, cfg4 eager / lazy, cfg5) and reports them
under "secondary" — value, ms per sweep and the roofline of their dominant
kernels from HIP events — so that one driver run carries the HBM-bound numbers
too (`--no-secondary` skips them).

With N > 1 (one rank per GPU) the headline stays the metric's own workload, cfg2, data-parallel
the way north_star asks: every rank owns its own 4096 independent batch rows A_r (per-GPU work fixed:
"scaling": "weak"), B is the replicated parameter, and B.grad is summed over the ranks by ONE RCCL
all-reduce per sweep, sent in row panels that leave while the weight-gradient GEMM is still running.
`value` = N x sweeps/s, so value(N) / (N x value(1)) is the scaling efficiency against the driver's N = 1
run; `single_gpu_value` is the same sweep timed on ONE GPU in the same job (no collective), and
`tensors_per_s` = 4096 x value (batch rows through forward+backward per second). After the headline the
same job runs BASELINE's configs[3] (cfg4: ONE global batch of 8192 rows split over the ranks, strong
scaling, bucket [W.grad || b.grad] all-reduced) and reports it as flat keys `cfg4_strong_*` together with
its own single-GPU figure; `--workload cfg4` makes it the headline instead.

`python bench.py --gpus N` with N > 1 and no rank environment starts its own N ranks
(one child process per GPU, before anything in THIS process touches the GPU), forwards
rank 0's line and exits non-zero if any rank fails or the job exceeds --rank-timeout.
At N > 1 the gradient all-reduce must run on the library's own RCCL communicator
("rccl-direct"); anything else is an error unless --allow-torch-comm is given.

ONE JSON line on stdout (rank 0), under 2 KB: `roofline` prices the dominant kernel from
HIP events attached to the kernel's own dispatch inside the timed region and carries, as
FLAT scalar keys, the figures of the other BASELINE configs (cfgN_*); `cpu_baseline` times
the NumPy oracle (the reference's arithmetic) on this host's cores for every config. The
full per-kernel detail goes to bench_detail.json (next to this file) and to stderr.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix, dense
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec
MIB = 1 << 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="default: cfg2 on one GPU (BASELINE's headline), cfg4 on several (BASELINE's batch-sharded config)")
    ap.add_argument("--size", type=int, default=0, help="override the problem size (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the headline workload")
    ap.add_argument("--force-secondary", action="store_true", help="run the other configs even with --size (tests: toy sizes on the CPU double)")
    ap.add_argument("--lazy", action="store_true",
                    help="opt-in lazy fusion of elementwise chains (minidiff_amd/lazy.py); default is eager")
    ap.add_argument("--graph", action="store_true",
                    help="capture one sweep into a hipGraph after warm-up and time K replays of it (N=1 only)")
    ap.add_argument("--comm", default=os.environ.get("MDHIP_COMM", "rccl"), choices=["rccl", "torch"])
    ap.add_argument("--allow-torch-comm", action="store_true",
                    help="N > 1: accept torch.distributed's RCCL backend when the library's own communicator cannot be built "
                         "(default: that is an error — the panelled, overlapped all-reduce would not be what ran)")
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="self-spawned N > 1 job: seconds before the ranks are killed")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full per-kernel detail is written")
    return ap.parse_args()


def spawn_ranks(args, entry):
    """`python bench.py --gpus N` without a launcher: start N copies of `entry` (one per GPU) with the rank environment
    torch.distributed.run would set, forward rank 0's JSON line, and fail loudly — every rank killed, non-zero exit — if
    one rank dies or the job exceeds --rank-timeout. Runs BEFORE anything in this process touches the GPU and never
    replaces a process (no os.exec*): the parent only waits."""
    import signal
    import socket
    import subprocess
    import threading

    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, out0 = [], []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, entry] + sys.argv[1:], env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr, stderr=None, text=(rank == 0)))

    def pump():
        for ln in procs[0].stdout:
            out0.append(ln)

    reader = threading.Thread(target=pump, daemon=True)
    reader.start()

    def kill_all():
        for sig in (signal.SIGTERM, signal.SIGKILL):
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, sig)      # (start_new_session: the child leads its own group — its helpers go with it)
                    except (ProcessLookupError, PermissionError):
                        pass
            t_end = time.time() + 5.0
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)

    deadline = time.time() + args.rank_timeout
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                failed = f"no result within --rank-timeout {args.rank_timeout:.0f} s"
                break
            time.sleep(0.1)
    except BaseException as e:   # (Ctrl-C, a signal from the driver: do not leave ranks behind)
        failed = f"interrupted ({type(e).__name__})"
    if failed:
        kill_all()
        print(f"[bench] {n}-rank job FAILED: {failed}; all ranks killed", file=sys.stderr)
        return 124 if failed.startswith("no result") else 1
    reader.join(timeout=10.0)
    lines = [ln for ln in out0 if ln.startswith("{")]
    if len(lines) != 1:
        print(f"[bench] {n}-rank job printed {len(lines)} JSON lines (expected 1)", file=sys.stderr)
        return 1
    sys.stdout.write(lines[0] if lines[0].endswith("\n") else lines[0] + "\n")
    sys.stdout.flush()
    return 0


class KernelTimer:
    """Brackets chosen backend calls with HIP events on libmdhip's stream."""

    def __init__(self, lib, capacity=8192):
        import ctypes as C
        self.C, self.lib = C, lib
        self.pool, self.used, self.enabled = [], [], False
        self.capacity = capacity
        self.only = None  # when set: bracket these tags only (every bracket costs ~4 us of stream time)
        self.attach = os.environ.get("MDHIP_BENCH_ATTACH", "1") != "0"   # GEMM calls: kernel-attached events instead of markers
        self.attach_tags = ("gemm_exec",)   # lazy mode: the call that materialises a deferred product
        self.marker_tags = set()            # tags whose call launched no attachable kernel: bracketed by marker events instead
        self._open = None                   # (start, stop) of the attach bracket that is open right now (brackets nest)

    def _event(self):
        if self.pool:
            return self.pool.pop()
        ev = self.C.c_void_p()
        self.lib.event_create(self.C.byref(ev))
        return ev

    def wrap(self, fn, tag):
        """`tag`: a name, or a function of the call's arguments returning one."""
        def timed(*a, **kw):
            if not self.enabled or len(self.used) >= self.capacity:
                return fn(*a, **kw)
            name = tag(*a, **kw) if callable(tag) else tag   # (before the call: it may change what the tag looks at)
            if self.only is not None and name not in self.only and not (name.startswith("matmul") and "matmul" in self.only):
                return fn(*a, **kw)
            e0, e1 = self._event(), self._event()
            if self.attach and name not in self.marker_tags:
                # the main kernel this call launches carries the two timestamps itself (mdhip_event_attach_next -> MD_LAUNCH /
                # the GEMM launchers / the fused-kernel launcher): the kernel's own duration, no marker packets in the stream
                # (a marker pair costs ~4 us of stream time and read 5-15 % low on the 25-50 us streaming kernels)
                # (brackets nest in lazy mode: materialising a deferred product first materialises its pending operand. The
                # enclosing bracket's events are set aside while the inner call runs and re-attached behind it, so that they
                # ride on the enclosing call's OWN kernel and not on the first launch that comes along)
                outer, outer_pending = self._open, self.C.c_int(0)
                if outer is not None:
                    self.lib.event_attach_cancel(self.C.byref(outer_pending))
                self.lib.event_attach_next(e0, e1)
                self._open = (e0, e1)
                try:
                    out = fn(*a, **kw)
                finally:
                    pending = self.C.c_int(0)
                    self.lib.event_attach_cancel(self.C.byref(pending))
                    self._open = outer
                    if outer is not None and outer_pending.value:
                        self.lib.event_attach_next(*outer)
                if pending.value:          # the call launched no attachable kernel (deferred product, generic path): nothing recorded;
                    self.pool += [e0, e1]  # bracket this tag with markers from now on — except the products, which may just be deferred
                    if not (name.startswith("matmul") or name in self.attach_tags):
                        self.marker_tags.add(name)
                else:
                    self.used.append((name, e0, e1))
                return out
            self.lib.event_record(e0)
            out = fn(*a, **kw)
            self.lib.event_record(e1)
            self.used.append((name, e0, e1))
            return out
        return timed

    def empty_bracket_ms(self, reps=20):
        """Elapsed time of a bracket with NOTHING inside: what the two event markers themselves add to every
        bracketed duration (a few microseconds: matters for the 25-50 us streaming kernels, not for the GEMMs)."""
        ms, tot = self.C.c_float(), 0.0
        for _ in range(reps):
            e0, e1 = self._event(), self._event()
            self.lib.event_record(e0)
            self.lib.event_record(e1)
            self.lib.sync()
            self.lib.event_elapsed_ms(e0, e1, self.C.byref(ms))
            tot += float(ms.value)
            self.pool += [e0, e1]
        return tot / reps

    def collect(self):
        ms = self.C.c_float()
        out = {}
        for tag, e0, e1 in self.used:
            self.lib.event_elapsed_ms(e0, e1, self.C.byref(ms))
            out.setdefault(tag, []).append(float(ms.value))
            self.pool += [e0, e1]
        self.used = []
        return out


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except Exception:
        return {}


def pmc_mean(pmc, prefix):
    vals = [v["hbm_bytes_per_launch"] for k, v in pmc.get("kernels", {}).items() if k.startswith(prefix) and not k.endswith("]")]
    return sum(vals) / len(vals) if vals else None


def cpu_baseline(workload, size):
    """The oracle (NumPy table + same tape) on the host cores: bounded sample."""
    from minidiff_amd.tape import build_engine
    from minidiff_amd import workloads
    from oracle.numpy_table import NumpyOracleTable

    md = build_engine(NumpyOracleTable, "oracle")
    kw = {}
    sample = ""
    scale = 1.0
    if workload in ("cfg2", "cfg5"):
        if size:
            kw["n"] = size
        sample = f"full {workload} sweep, best of 5 after 1 warm-up"
        reps = 5
    elif workload == "cfg3":
        n = size or 100_000_000
        kw["n"] = min(n, 10_000_000)
        scale = kw["n"] / n
        sample = f"N={kw['n']} (1/{int(round(1/scale))} of the workload, time scaled linearly), best of 3 after 1 warm-up"
        reps = 3
    else:
        if size:
            kw["batch"] = size
        if os.environ.get("MDHIP_BENCH_CFG4_DIM"):   # tests only (as in run())
            kw["d_in"] = kw["d_out"] = int(os.environ["MDHIP_BENCH_CFG4_DIM"])
        sample = "full cfg4 sweep (global batch on one host), best of 3 after 1 warm-up"
        reps = 3
    state, step = workloads.MAKERS[workload](md, **kw)
    step()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        best = min(best, time.perf_counter() - t0)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # the oracle's gradients of that sweep (same seeds as the device run): the metric's second half, "grad L-inf vs NumPy"
    grads = {name: np.asarray(state[name].grad.as_numpy()) for name in GRAD_NAMES[workload]}
    if workload == "cfg4":
        # the relu decision of every (row, column) as the NumPy engine takes it, and max|X|: what a mask flip can move (see grad_linf_rel)
        Xh, Wh, bh = (np.asarray(state[k].as_numpy()) for k in ("X", "W", "b"))
        grads["_mask"] = (Xh @ Wh + bh) > 0
        grads["_xmax"] = float(np.abs(Xh).max())
    return {
        "value": scale / best, "unit": "passes/s", "cores": cores, "kind": "port",
        "sample": sample + f"; numpy {np.__version__}; GEMM all cores, ufuncs 1 thread",
    }, grads


GRAD_NAMES = {"cfg2": ("A", "B"), "cfg3": ("x", "y"), "cfg4": ("W", "b"), "cfg5": ("A", "B")}


def grad_linf_rel(device_grads, oracle_grads):
    """norm-wise max|g_hip - g_np| / max|g_np| per gradient (the oracle may hold a prefix of the device's
    vector: cfg3's CPU sample is a tenth of the workload, and element i of its gradients depends on x[i], y[i] alone).

    cfg4 is discontinuous: W.grad = X^T M and b.grad = sum_rows M with M = (X @ W + b > 0), and the device's product sums in
    another order than OpenBLAS, so pre-activations within rounding of zero (a few tens of 33.5 M) take the other branch. A
    flipped M[i, j] moves column j of W.grad by row i of X — up to max|X| per element, ~1e-2 of max|W.grad| — which says nothing
    about the arithmetic. The comparison therefore allows each column what its flips can account for:
        err[k, j] = max(0, |dev - ref|[k, j] - flips[j] * max|X|)        (b.grad: flips[j] * 1)
    with flips[j] COUNTED from the device's own mask (recomputed on the device after the timed region) against the NumPy
    engine's. Reported: the adjusted figure per gradient, the raw one (`*_raw`) and the number of flips (`mask_flips`)."""
    out = {}
    flips = None
    if "_mask" in oracle_grads and "_mask" in device_grads:
        diff = device_grads["_mask"] != oracle_grads["_mask"]
        flips = diff.sum(axis=0).astype(np.float64)
        out["mask_flips"] = int(diff.sum())
        out["mask_entries"] = int(diff.size)
    for name, ref in oracle_grads.items():
        dev = device_grads.get(name)
        if dev is None or name.startswith("_"):
            continue
        if flips is not None:
            e = np.abs(dev.astype(np.float64) - ref.astype(np.float64))
            scale = max(float(np.max(np.abs(ref))), 1e-300)
            out[name + "_raw"] = float(e.max() / scale)
            allow = flips * (oracle_grads["_xmax"] if dev.ndim == 2 else 1.0)
            out[name] = float(np.maximum(e - allow, 0.0).max() / scale)
            continue
        d = dev.reshape(-1)[: ref.size] if dev.size != ref.size else dev.reshape(-1)
        r = ref.reshape(-1).astype(np.float64)
        out[name] = float(np.max(np.abs(d.astype(np.float64) - r)) / max(np.max(np.abs(r)), 1e-300))
    return out


PREROLL = {"cfg2": 12, "cfg3": 20, "cfg4": 10, "cfg5": 40}
DEFAULT_SIZE = {"cfg2": 4096, "cfg3": 100_000_000, "cfg4": 8192, "cfg5": 2048}


def describe(workload, n, lazy):
    mode = " [lazy fusion]" if lazy else ""
    return {
        "cfg2": f"cfg2: C=A@B ({n}x{n} fp32), C.backward(); 3 GEMMs NN/NT/TN" + mode,
        "cfg3": f"cfg3: sum((sin(x)*y)**2).backward(), N={n} fp32" + (mode if lazy else " [eager: 11 kernels]"),
        "cfg4": f"cfg4: sum(relu(X@W+b)).backward(), batch {n} x 4096 -> 4096, row-sharded" + mode,
        "cfg5": f"cfg5: second order on {n}x{n} matmul, 5 GEMMs" + (" [lazy: unread products are never launched - NOT the 5-GEMM workload]" if lazy else ""),
    }[workload]


def _mean(v):
    return sum(v) / len(v) if v else None


def _hbm(kernel, nbytes, ms, durs=None, **extra):
    ach = nbytes / (ms * 1e-3) / 1e9
    d = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
         "traffic": None, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms}
    if durs:
        d.update(_stats(durs))
    d.update(extra)
    return d


def _stats(v):
    """mean / min / median of a list of launch durations (ms)."""
    if not v:
        return {}
    w = sorted(v)
    return {"avg_launch_ms": sum(w) / len(w), "min_launch_ms": w[0], "median_launch_ms": w[len(w) // 2]}


def rooflines(workload, lazy, state, kernel_ms, ms_per_step, size, world, pmc, n_bracketed=0, dma=True):
    """-> (roofline of the dominant kernel, extra per-kernel detail) from the HIP-event durations of the timed region."""
    per_sweep = kernel_ms.pop("_per_sweep", None)
    detail = {k: {"launches": len(v), "avg_ms": _mean(v)} for k, v in sorted(kernel_ms.items())}
    if workload in ("cfg2", "cfg5") or (workload == "cfg4"):
        durs = [d for k, v in kernel_ms.items() if k.startswith("matmul") for d in v]
        n_gemm = {"cfg2": 3, "cfg4": 2, "cfg5": 5}[workload]
        flop_per_launch = state["flops"] / n_gemm
        main = None
        # in lazy mode the weight-gradient matmul call also launches the fused pass that produces its operand:
        # price the GEMM on the calls that contain nothing else
        # (16-B aligned operands and whole tiles — every BASELINE shape — run the direct-to-LDS kernels; the option gemm_glds = 0
        # (experiments) or a ragged shape the register-staged k_gemm_f32_mfma)
        gemm_kernel = "k_gemm_f32_kc_glds (NN,NT) / k_gemm_f32_tn_glds (TN)" if dma else "k_gemm_f32_mfma"
        if workload == "cfg4" and lazy:
            # lazy mode defers the products: the forward GEMM runs inside the loss reduction with the bias / relu / sum
            # epilogue (bracket "sum_all": one launch, the last block sums the per-block relu sums), the weight-gradient GEMM when W.grad is materialised
            durs = kernel_ms.get("sum_all", []) + kernel_ms.get("gemm_exec", [])
            gemm_kernel = ("k_gemm_f32_kc_glds (NN with the bias + relu-sum + mask epilogue) and k_gemm_f32_tn_glds (TN)" if dma
                           else "k_gemm_f32_mfma (NN with the bias + relu-sum + mask epilogue) and k_gemm_f32_mfma (TN)")
        elif lazy:
            durs = kernel_ms.get("gemm_exec", [])
        elif workload == "cfg4":
            durs = kernel_ms.get("matmul_nn", []) + kernel_ms.get("matmul_tn", [])
        if durs:
            avg = _mean(durs)
            ach = flop_per_launch / (avg * 1e-3) / 1e12
            if n_bracketed and len(durs) != n_gemm * n_bracketed:
                # N > 1: the weight gradient is produced in row panels — more, smaller GEMM launches than the plain sweep's.
                # Price the GEMM work of the bracketed sweeps as a whole: their flops over the sum of the launches' durations.
                flop_per_launch = state["flops"] * n_bracketed / len(durs)
                ach = state["flops"] * n_bracketed / (sum(durs) * 1e-3) / 1e12
            committed = pmc_mean(pmc, "k_gemm_f32_") if workload == "cfg2" and not size else None
            main = {"bound": "mfma", "kernel": gemm_kernel, "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK_TFLOPS,
                    "traffic": None,  # not measured in this run (PMC passes need the profiler)
                    "traffic_committed": committed, "traffic_committed_source": pmc.get("_source") if committed else None,
                    "launches": len(durs), **_stats(durs), "flop_per_launch": flop_per_launch,
                    "algorithmic_bytes_per_launch": 3 * 4 * (size or 4096) ** 2 if workload == "cfg2" else None}
        if workload == "cfg4":
            rows, cols = state["rows"], 4096
            e = rows * cols
            tail = {}
            if not lazy:
                # SURVEY 8d byte counts per call as issued by the untouched tape (stride-0 / scalar operands = 0)
                for tag, name, nbytes in (("add", "bias add", 8 * e + 4 * cols), ("greater", "z > 0 -> bool", 5 * e),
                                          ("where", "where(mask, z, 0)", 9 * e), ("sum_all", "sum (loss)", 4 * e),
                                          ("multiply", "mask product g*mask (f32 x bool)", 5 * e),
                                          ("sum_cols", "column sum (bias gradient, reduce-to-shape)", 4 * e + 4 * cols)):
                    if kernel_ms.get(tag):
                        tail[tag] = _hbm(name, nbytes, _mean(kernel_ms[tag]), durs=kernel_ms[tag])
                if "multiply" in tail and "sum_cols" in tail:
                    ms = tail["multiply"]["avg_launch_ms"] + tail["sum_cols"]["avg_launch_ms"]
                    tail["backward_pair"] = _hbm("elementwise + reduce-to-shape backward: mask product (k_ew_fast), then column sum (k_reduce_cols_strips): one launch each",
                                                 9 * e + 4 * cols, ms, note="north_star's >= 60 % HBM target is on this pair")
            else:
                # the forward tail (bias add, relu, mask, loss) lives in the NN GEMM's epilogue: no kernel of its own
                if kernel_ms.get("materialize"):
                    tail["backward_pair"] = _hbm("k_fused_evalcols: g*mask written AND column-summed in one pass over the mask, ONE launch (the last block of a strip merges its partial rows)",
                                                 5 * e + 8 * cols, _mean(kernel_ms["materialize"]),
                                                 eager_algorithmic_bytes=9 * e + 4 * cols,
                                                 note="fused bytes: the bool mask read once, g*mask written once; the eager figure is not mixed in")
            detail["hbm_tail"] = tail
        return main, detail
    n = state["rows"]
    if lazy:
        # fused: one reduce pass over (x, y) for the loss (8N), one two-output pass for both gradients (reads x, y
        # once, writes x.grad and y.grad: 16N) = SURVEY 8d's fused lower bound
        fused_bytes = 24 * n
        ach = fused_bytes / (ms_per_step * 1e-3) / 1e9
        main = {"bound": "hbm", "kernel": "k_fused_redall + k_fused_eval2 (run-time specialised), whole sweep",
                "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                "fused_algorithmic_bytes_per_sweep": fused_bytes, "eager_algorithmic_bytes_per_sweep": state["bytes"],
                "note": "achieved = fused bytes / whole-sweep wall time (2 launches); the eager figure (100N bytes) is not mixed in"}
        k = {}
        if kernel_ms.get("sum_all"):
            k["loss_pass"] = _hbm("k_fused_redall: sum((sin(x)*y)**2), reads x and y; one launch (the last block sums the block partials)", 8 * n, _mean(kernel_ms["sum_all"]))
        if kernel_ms.get("materialize_many"):
            k["gradient_pass"] = _hbm("k_fused_eval2: x.grad and y.grad in one pass, one sincos per element", 16 * n, _mean(kernel_ms["materialize_many"]))
        detail["fused_kernels"] = k
        return main, detail
    # dominant kernel of the eager chain: the f32 x f32 multiply (5 of the 11 launches per sweep: read 2 x 4N, write 4N)
    durs = kernel_ms.get("multiply", [])
    main = None
    if durs:
        avg = _mean(durs)
        # whole sweep: every bracketed call of one sweep (detail pass), launches x mean duration
        tot_ms = sum(c * m for c, m in (per_sweep or {}).values()) or sum(sum(v) for v in kernel_ms.values())
        steps = 1 if per_sweep else (len(kernel_ms.get("sin", [])) or 1)
        committed = pmc_mean(pmc, "k_ew_fast<BinaryBody<BMul, float, float, float, float, 1, 1, true>") if not size else None
        main = _hbm("k_ew_fast<BinaryBody<BMul,f32,f32>> (multiply, both operands streamed)", 12 * n, avg, durs=durs, launches=len(durs),
                    traffic_committed=committed, traffic_committed_source=pmc.get("_source") if committed else None,
                    whole_sweep={"algorithmic_bytes": state["bytes"], "kernel_ms": tot_ms / steps,
                                 "GB/s": state["bytes"] * steps / (tot_ms * 1e-3) / 1e9,
                                 "frac": state["bytes"] * steps / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    chain = {}
    for tag, nbytes in (("sin", 8 * n), ("cos", 8 * n), ("power", 8 * n), ("sum_all", 4 * n), ("multiply_scalar", 4 * n)):
        if kernel_ms.get(tag):
            chain[tag] = _hbm(tag, nbytes, _mean(kernel_ms[tag]), durs=kernel_ms[tag])
    detail["chain"] = chain
    return main, detail


def _r(x, sig=5):
    """Numbers in the stdout line carry `sig` significant digits (the line stays under 2 KB; bench_detail.json has them all)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{sig}g}")


def flat_keys(results):
    """The other BASELINE configs as FLAT scalar keys for `roofline` (the driver's record keeps scalars of `roofline`, `config`
    and `cpu_baseline` and only the NAMES of anything nested or top-level): passes/s per config, and for every kernel a
    verdict prices — fraction of peak (157.3 TFLOP/s for GEMM kernels, 8 TB/s of ALGORITHMIC bytes, SURVEY 8d, for streaming
    kernels), mean duration in us (the kernel's own dispatch timestamps) and the bytes / flop it is priced on, so that every
    fraction can be recomputed from the record alone. `results`: {"cfg3": run result, "cfg3_lazy": .., "cfg4": .., ..}."""
    def pick(d, *path):
        for k in path:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return d

    f = {}

    def put(key, v):
        if v is not None:
            f[key] = v

    def kern(prefix, r, us=True, nbytes=False):
        if not r:
            return
        put(prefix + "_frac", r.get("frac"))
        if us and r.get("avg_launch_ms") is not None:
            put(prefix + "_us", r["avg_launch_ms"] * 1e3)
        if nbytes:
            put(prefix + "_bytes", r.get("algorithmic_bytes_per_launch"))

    for name, r in results.items():
        if not r or "error" in r:
            f[name + "_error"] = 1
            continue
        put(name + "_passes_per_s", r["value"])
        roof = r.get("roofline") or {}
        if name == "cfg3":
            kern("cfg3_mul", roof, nbytes=True)
            put("cfg3_whole_sweep_frac", pick(roof, "whole_sweep", "frac"))
            chain = pick(r, "kernels", "chain") or {}
            kern("cfg3_sin", chain.get("sin"))
            kern("cfg3_cos", chain.get("cos"), us=False)
            kern("cfg3_loss_sum", chain.get("sum_all"), us=False)
        elif name == "cfg3_lazy":
            put("cfg3_lazy_frac", roof.get("frac"))                     # whole sweep: 24N bytes / wall time
            fused = pick(r, "kernels", "fused_kernels") or {}
            kern("cfg3_lazy_loss_pass", fused.get("loss_pass"), us=False)
            kern("cfg3_lazy_grad_pass", fused.get("gradient_pass"), us=False)
        elif name in ("cfg4", "cfg4_lazy"):
            put(name + "_gemm_frac", roof.get("frac"))
            tail = pick(r, "kernels", "hbm_tail") or {}
            kern(name + "_pair", tail.get("backward_pair"), nbytes=True)
            if name == "cfg4":
                kern("cfg4_loss_sum", tail.get("sum_all"))
                kern("cfg4_colsum", tail.get("sum_cols"))
                kern("cfg4_maskprod", tail.get("multiply"))
                kern("cfg4_bias_add", tail.get("add"), us=False)
                kern("cfg4_where", tail.get("where"), us=False)
                kern("cfg4_greater", tail.get("greater"), us=False)
        elif name in ("cfg5", "cfg5_graph", "cfg2", "cfg2_weak", "cfg4_strong"):
            put(name + "_frac", roof.get("frac"))
            if roof.get("avg_launch_ms") is not None:
                put(name + "_gemm_us", roof["avg_launch_ms"] * 1e3)
            if name in ("cfg2_weak", "cfg4_strong"):    # N > 1 only: what the scaling of this config is judged by
                put(name + "_ms_per_step", r.get("ms_per_step"))
                put(name + "_tensors_per_s", r.get("tensors_per_s"))
                put(name + "_single_gpu_passes_per_s", pick(r, "single_gpu_same_workload", "value"))
                put(name + "_allreduce_alone_ms", pick(r, "config", "allreduce_alone_ms"))
                put(name + "_allreduce_busbw_GBps", pick(r, "config", "allreduce_busbw_GBps"))
    return f


# sweeps per secondary config: ~40 ms of timed region each (ms per sweep: 1.7 / 0.4 / 3.9 / 3.7 / 0.65 / 0.63)
SECONDARY_STEPS = {"cfg3": 24, "cfg3_lazy": 100, "cfg4": 10, "cfg4_lazy": 10, "cfg5": 60, "cfg5_graph": 60}

# keys of `roofline` given up first when the line would pass 2 KB (everything stays in bench_detail.json)
DROP_ORDER = ["cfg5_graph_gemm_us", "cfg5_graph_frac", "cfg4_greater_frac", "cfg4_where_frac", "cfg3_cos_frac", "cfg3_loss_sum_frac", "traffic_committed_source", "min_launch_ms",
              "cfg3_lazy_loss_pass_frac", "cfg3_lazy_grad_pass_frac", "cfg4_lazy_gemm_frac", "cfg4_bias_add_frac", "cfg3_mul_bytes",
              "cfg4_lazy_pair_bytes", "launches", "median_launch_ms", "cfg3_sin_us", "cfg4_maskprod_us", "cfg4_colsum_us"]


def main(entry=None):
    """`entry`: the script a self-spawned rank runs (default: this file; tests pass their own wrapper)."""
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # no launcher around us: be the launcher (nothing in this process has touched the GPU yet, and nothing will)
        raise SystemExit(spawn_ranks(args, entry or os.path.abspath(__file__)))
    # Only the JSON line may reach stdout: libraries print banners there (RCCL's version block, gloo's "connected to N peer
    # ranks"). File descriptor 1 points at stderr until the line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs", file=sys.stderr)
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MDHIP_DEVICE"] = str(local_rank)

    # MDHIP_BENCH_FORCE_DIST=1 runs the N>1 code path (process group, ncclUniqueId exchange, RCCL
    # communicator, per-sweep all-reduce, max-over-ranks timing) at world size 1 — the only way to
    # rehearse it on a one-GPU box
    force_dist = os.environ.get("MDHIP_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    WORLD, RANK, USE_DIST = world, rank, use_dist   # (run() shadows the lower-case names in its solo mode)
    if args.workload is None:
        args.workload = "cfg2"      # at every N: the workload BASELINE's metric is quoted on (N > 1: weak scaling over batch rows)
    dist = torch = None
    if use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if torch.cuda.is_available():  # device_count() alone does not touch the GPU; set_device binds torch to OUR card
            torch.cuda.set_device(local_rank)

    from minidiff_amd import _capi, workloads, dp
    from minidiff_amd import ndarray as nd
    from minidiff_amd.hip_backend import HipBackendTable
    from minidiff_amd.tape import build_engine

    lib = _capi.load()  # ImportError if the HIP extension is missing: no fallback
    timer = KernelTimer(lib)

    # instrumented copy of the table: same functions, the kernels of the BASELINE graphs bracketed by events
    def is_arr(x):
        return isinstance(x, nd.DeviceArray)

    def tag_matmul(a, b, *_, **__):
        if is_arr(a) and is_arr(b) and a.ndim == 2 and b.ndim == 2:
            at, bt = a._strides[-1] != 1, b._strides[-1] != 1
            return "matmul_" + ("t" if at else "n") + ("t" if bt else "n")
        return "matmul"

    tags = {
        "matmul": tag_matmul,
        # (cfg3: the five full-array products vs the scaled stride-0 seed; cfg4: the one mask product)
        "multiply": lambda a, b, *_, **__: "multiply" if is_arr(a) and is_arr(b) else "multiply_scalar",
        "sum": lambda a, axis=None, *_, **__: "sum_all" if axis is None or axis == () else "sum_cols",
        "sin": "sin", "cos": "cos", "power": "power", "add": "add", "greater": "greater", "where": "where",
    }
    ns = {k: v for k, v in vars(HipBackendTable).items() if not k.startswith("__")}
    for name, tag in tags.items():
        ns[name] = staticmethod(timer.wrap(getattr(HipBackendTable, name), tag))
    ns["_materialize_many"] = staticmethod(timer.wrap(nd.materialize_many, "materialize_many"))
    Table = type("HipBackendTableTimed", (), ns)
    md = build_engine(Table, "hip")
    # lazy mode launches the fused pass of a pending operand when something needs it in memory
    _plain_materialize = nd.DeviceArray._materialize
    from minidiff_amd import lazy as _lz

    def tag_materialize(arr):
        e = arr._expr
        return "gemm_exec" if e is not None and e.kind == _lz.GEMM else "materialize"   # a deferred product vs a fused elementwise pass

    nd.DeviceArray._materialize = timer.wrap(_plain_materialize, tag_materialize)

    pmc = load_pmc_traffic()

    def barrier():
        lib.sync()
        if use_dist:
            dist.barrier()

    def make_comm(workload):
        comm, kind = None, "none"
        if use_dist and workload in ("cfg2", "cfg4") and os.environ.get("MDHIP_BENCH_HOST_COMM") == "1":
            # tests only (tests/test_bench_contract.py): the N > 1 control flow over gloo on the CPU test double
            return dp.HostComm(rank, world, dist, torch), "gloo-host(test)"
        if use_dist and workload in ("cfg2", "cfg4"):
            if args.comm == "rccl":
                err = None
                try:
                    comm = dp.RcclComm(rank, world, dist)
                except Exception as e:  # communicator could not be built (on this or another rank: RcclComm fails on all together)
                    err = e
                # every rank must take the same path: agree over the gloo control plane
                flags = [None] * world
                dist.all_gather_object(flags, err is None)
                if all(flags):
                    kind = "rccl-direct"
                else:
                    if comm is not None:
                        comm.close()
                        comm = None
                    why = f"[rank {rank}] direct RCCL communicator unavailable ({err or 'failed on another rank'})"
                    if not args.allow_torch_comm:
                        # every rank takes this branch together (the flags were all-gathered): the job ends with rc != 0 instead
                        # of a line measured on a host-synchronous fallback that nobody asked for
                        print(why + "; refusing to fall back (pass --allow-torch-comm to accept torch.distributed's backend)", file=sys.stderr)
                        raise SystemExit(3)
                    print(why + "; using torch.distributed nccl (--allow-torch-comm)", file=sys.stderr)
            elif not args.allow_torch_comm and world > 1:
                print("[bench] --comm torch at N > 1 needs --allow-torch-comm (the product path is the library's own RCCL communicator)", file=sys.stderr)
                raise SystemExit(3)
            if comm is None:  # same data path through torch's RCCL
                comm = dp.TorchComm(rank, world, dist, torch)
                kind = "rccl-torch"
        return comm, kind

    def rccl_ranks(comm):
        """What ncclCommCount says about the live communicator (None: not the direct RCCL path)."""
        if not isinstance(comm, dp.RcclComm):
            return None
        import ctypes as C
        n = C.c_int(0)
        lib.comm_count(C.byref(n))
        return int(n.value)

    def run(workload, lazy, steps, warmup, graph=False, size=0, keep=None, solo=False, want_grads=False):
        """One workload: pre-roll, W warm-up sweeps, K timed sweeps between barriers, max over ranks.
        `keep`: dict carrying (state, step) between the eager and the lazy run of one workload.
        `solo` (N > 1 jobs): every rank runs the UN-sharded workload on its own GPU with no collective, and the time is the
        calling rank's own — the same-job N = 1 figure of the workload the N > 1 headline is quoted on."""
        world, rank, use_dist = (1, 0, False) if solo else (WORLD, RANK, USE_DIST)
        prev_lazy = nd.set_lazy(bool(lazy))
        kw = {}
        if workload == "cfg2":
            kw = {"rank": rank}
            if size:
                kw["n"] = size
        elif workload == "cfg4":
            kw = {"rank": rank, "world": world}
            if size:
                kw["batch"] = size
            if os.environ.get("MDHIP_BENCH_CFG4_DIM"):   # tests only: a layer the CPU double's plain-loop GEMM can afford
                kw["d_in"] = kw["d_out"] = int(os.environ["MDHIP_BENCH_CFG4_DIM"])
        elif size:
            kw = {"n": size}
        if keep is not None and "state" in keep:
            state, step = keep["state"], keep["step"]
        else:
            state, step = workloads.MAKERS[workload](md, **kw)
            if keep is not None:
                keep["state"], keep["step"] = state, step
        comm, comm_kind = (None, "none") if solo else make_comm(workload)
        n_rccl = rccl_ranks(comm)
        if n_rccl is not None and n_rccl != world:
            raise SystemExit(f"[bench] the RCCL communicator reports {n_rccl} ranks, the job has {world}")
        sync = dp.GradSync(md, state["params"] if workload == "cfg4" else state["params"][:1], comm, force=force_dist,
                           overlap=os.environ.get("MDHIP_DP_OVERLAP", "1") != "0")

        def sweep():
            step()
            sync()

        # untimed pre-roll before the W warm-up sweeps: the first ~15 ms of work after process start run
        # at ramping clocks and grow the allocator cache (W <= 2 alone measured 7 % low on cfg2)
        # (a fixed count, not a time: every rank must issue the same number of collectives)
        preroll = PREROLL[workload]
        for _ in range(preroll):
            sweep()
        lib.sync()
        captured = None
        # N > 1: where the sweep's kernels are short (cfg4 at 8 ranks: 65-250 us) ~7 us of Python dispatch per backend call would
        # sit between them: the sweep can be replayed as hipGraph SEGMENTS with the collectives between them (graph.SegmentedSweep).
        # Where they are long (cfg2: 0.24-0.9 ms) the host runs ahead anyway and every segment cut costs ~14 us of graph start-up.
        # MDHIP_DP_GRAPH: "0" eager sweeps, "1" segments, unset = a three-sweep trial of both before the timed region, the faster
        # one (max over ranks, so every rank takes the same) runs the K timed sweeps. A build that cannot capture stays eager.
        dp_graph = os.environ.get("MDHIP_DP_GRAPH", "auto")
        trial = None
        segmented = bool(use_dist and comm is not None and sync.active and not graph and dp_graph != "0")
        if segmented:
            from minidiff_amd.graph import can_capture
            segmented = can_capture(lib)
        if segmented:
            sweep()
            timer.enabled = True   # per-kernel durations for the roofline come from the eager warm-up sweeps
            for _ in range(max(warmup, 1)):
                sweep()
            lib.sync()
            timer.enabled = False
            from minidiff_amd.graph import SegmentedSweep
            try:
                captured = SegmentedSweep(sweep, comm)
                captured.replay()
                run_one = captured.replay
                if dp_graph != "1":
                    def trial_ms(fn, n=3):
                        fn()
                        lib.sync()
                        barrier()
                        t = time.perf_counter()
                        for _ in range(n):
                            fn()
                        lib.sync()
                        dt = (time.perf_counter() - t) / n * 1e3
                        if dist is not None and dist.is_initialized():
                            tt = torch.tensor([dt], dtype=torch.float64)
                            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                            dt = float(tt.item())
                        return dt
                    trial = {"segments_ms": trial_ms(captured.replay)}
                    # (the eager candidate as it would run in the timed region: its GEMM launches carry the roofline's events)
                    warm_events, timer.used = timer.used, []
                    timer.only, timer.enabled = {"matmul"}, True
                    trial["eager_ms"] = trial_ms(sweep)
                    timer.only, timer.enabled = None, False
                    timer.collect()        # (the trial's own brackets: recycled, not priced)
                    timer.used = warm_events
                    if trial["eager_ms"] <= trial["segments_ms"]:
                        timer.collect()    # (.. nor the warm-up's: the timed region's own launches price the roofline)
                        segs = {"segments": captured.segments, "collective_calls": captured.calls}
                        captured.close()
                        captured, segmented, run_one = None, False, sweep
                        trial.update(segs)
            except RuntimeError as e:
                captured, segmented = None, False
                try:        # the failed capture may have stopped inside backward(): put the gradient sync back to "between sweeps"
                    sync()
                except Exception:
                    pass
                lib.sync()
                if rank == 0:
                    print(f"[bench] segmented graph capture unavailable ({e}); eager sweeps", file=sys.stderr)
                run_one = sweep
        if segmented:
            pass
        elif graph:
            if use_dist:
                raise SystemExit("--graph is a single-GPU mode (N > 1 replays graph segments by default: MDHIP_DP_GRAPH)")
            # single kernels cannot be bracketed inside a replay: the per-kernel durations the roofline
            # needs come from the (eager, identical) warm-up sweeps instead
            sweep()  # cold start (code object load, allocator growth) stays out of the kernel averages
            timer.enabled = True
            for _ in range(max(warmup, 1)):
                sweep()
            lib.sync()
            timer.enabled = False
            from minidiff_amd.graph import CapturedSweep
            captured = CapturedSweep(step, warmup=0)
            captured.replay()
            run_one = captured.replay
        else:
            for _ in range(warmup):
                sweep()
            run_one = sweep
        barrier()
        # inside the timed region only the dominant kernel is bracketed (an empty bracket costs ~4 us of stream time: eleven
        # of them per sweep took 20 % off the fused cfg3 sweep); the per-kernel detail comes from a few extra sweeps afterwards
        dominant = {"cfg2": {"matmul"}, "cfg5": {"matmul", "gemm_exec"},
                    "cfg4": {"gemm_exec", "sum_all"} if lazy else {"matmul"},
                    "cfg3": set() if lazy else {"multiply"}}[workload]
        timer.only = dominant
        timer.enabled = not graph and not segmented and bool(dominant)
        t0 = time.perf_counter()
        for _ in range(steps):
            run_one()
        lib.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            dist.barrier()
        kernel_ms = timer.collect()
        ms_per_step = elapsed / steps * 1e3
        timer.only = None
        # SURVEY 8d also asks for the median and the minimum of single sweeps: ten more sweeps, each synchronised at its
        # end (outside the timed region: a per-sweep sync costs the overlap between consecutive sweeps)
        single = []
        if not graph:   # (eager sweeps also in segmented mode: what one sweep costs with Python dispatch in it)
            for _ in range(10):
                barrier()
                t1 = time.perf_counter()
                sweep()
                lib.sync()
                single.append((time.perf_counter() - t1) * 1e3)
            single.sort()
        if not graph and workload in ("cfg3", "cfg4"):   # detail pass: every kernel of the sweep bracketed, outside the timed region
            timer.enabled = True
            for _ in range(min(steps, 5)):
                sweep()
            lib.sync()
            timer.enabled = False
            detail_ms = timer.collect()
            n_detail = min(steps, 5)
            # launches per sweep and mean duration of every bracketed call, for the whole-sweep figures
            kernel_ms["_per_sweep"] = {k: (len(v) / n_detail, sum(v) / len(v)) for k, v in detail_ms.items()}
            for k, v in detail_ms.items():
                kernel_ms.setdefault(k, v)
            if use_dist:
                dist.barrier()

        # the collective alone (un-overlapped, outside the timed region): time and bus bandwidth
        allreduce_ms = busbw = None
        if comm is not None and sync.active:
            import ctypes as C
            grads = [p.grad for p in sync.params]
            buf = sync.bucket if sync.bucket is not None else grads[0]._data
            e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
            lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
            reps = 5
            comm.allreduce_sum_(buf)
            barrier()
            lib.event_record(e0)
            for _ in range(reps):
                comm.allreduce_sum_(buf)
            lib.event_record(e1)
            lib.sync()
            if comm_kind == "rccl-direct":
                lib.event_elapsed_ms(e0, e1, C.byref(ms))
                allreduce_ms = float(ms.value) / reps
            if allreduce_ms and world > 1:
                busbw = 2.0 * (world - 1) / world * sync.nbytes / (allreduce_ms * 1e-3) / 1e9
            barrier()
        # cfg2/cfg3/cfg5: every rank runs a full sweep on its own shard (weak); cfg4: one global batch (strong)
        if workload == "cfg4":
            value, scaling = steps / elapsed, "strong"
        else:
            value, scaling = world * steps / elapsed, "weak"
        import ctypes as _C
        glds = _C.c_int64(1)
        lib.debug_get_option(b"gemm_glds", _C.byref(glds))
        roof, detail = rooflines(workload, lazy, state, kernel_ms, ms_per_step, size, world, pmc,
                                 n_bracketed=(max(warmup, 1) if (graph or segmented) else steps), dma=bool(glds.value))
        n = size or DEFAULT_SIZE[workload]
        res = {
            "value": value, "unit": "passes/s", "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup, "preroll_sweeps": preroll,
            "scaling": scaling,
            "single_sweep_ms": {"median": single[len(single) // 2], "min": single[0], "n": len(single)} if single else None,
            "config": {"workload": describe(workload, n, lazy), "parallelism": f"dp{world}", "lazy_fusion": bool(lazy),
                       "graph_replay": bool(graph) or ({"segments": captured.segments, "collective_calls": captured.calls} if segmented else False),
                       "sweep_trial_eager_ms": trial["eager_ms"] if trial else None, "sweep_trial_segments_ms": trial["segments_ms"] if trial else None,
                       "collective": comm_kind, "rccl_ranks": n_rccl, "allreduce_bytes": sync.nbytes if use_dist else 0,
                       "allreduce_overlapped_sweeps": sync.overlapped, "allreduce_panels": getattr(sync, "panels", 1),
                       "allreduce_alone_ms": allreduce_ms, "allreduce_busbw_GBps": busbw},
            "roofline": roof, "kernels": detail,
        }
        if workload in ("cfg2", "cfg4"):
            # batch rows through forward+backward per second (cfg4: one global batch; cfg2: every rank's own rows — `value`
            # already counts all ranks' sweeps)
            res["tensors_per_s"] = (size or DEFAULT_SIZE[workload]) * value
        if want_grads:   # the last sweep's gradients, copied to the host AFTER the timed region (the L-inf check against the oracle)
            res["_grads"] = {name: np.asarray(state[name].grad.as_numpy()) for name in GRAD_NAMES[workload]}
            if workload == "cfg4" and world == 1:
                # the device's own relu decisions: the same product + bias add the sweep runs (eager kernels; the lazy epilogue's
                # mask agrees with it bit for bit, tests/test_lazy_fusion.py), compared with the NumPy engine's in grad_linf_rel
                was = nd.set_lazy(False)
                z = md.backend.add(md.backend.matmul(state["X"]._data, state["W"]._data), state["b"]._data)
                res["_grads"]["_mask"] = np.asarray(md.backend.as_numpy(md.backend.greater(z, 0)))
                del z
                nd.set_lazy(was)
        sync.close()
        if comm is not None:
            comm.close()
        if captured is not None:
            captured.close()
        nd.set_lazy(prev_lazy)
        return res

    check_grads = rank == 0 and world == 1 and not args.no_cpu_baseline
    head = run(args.workload, args.lazy, args.steps, args.warmup, graph=args.graph, size=args.size, want_grads=check_grads)
    head_grads = head.pop("_grads", None)
    event_overhead_ms = timer.empty_bracket_ms()
    # N > 1: the same job also times the headline workload on ONE GPU (every rank on its own card, no collective; rank 0's
    # figure) — the N = 1 value its scaling is judged against, from the same boxes and the same clocks
    def solo_run(workload, lazy):
        gc.collect()
        lib.empty_cache()
        r1 = run(workload, lazy, min(args.steps, 10), min(args.warmup, 2), size=args.size, solo=True)
        s = {"value": r1["value"], "unit": "passes/s", "ms_per_step": r1["ms_per_step"], "n_gpus": 1,
             "workload": r1["config"]["workload"],
             "note": "the workload un-sharded on ONE GPU, timed in this job (all ranks run it at once, no collective)"}
        if "tensors_per_s" in r1:
            s["tensors_per_s"] = r1["tensors_per_s"]
        gc.collect()
        lib.empty_cache()
        return s

    solo = solo_run(args.workload, args.lazy) if world > 1 and not args.graph else None

    # ---- CPU leg (rank 0, N = 1): the NumPy engine on the same seeds, once per WORKLOAD — its passes/s (the reported baseline)
    # and its gradients, against which every device run of that workload (eager and lazy) is compared norm-wise
    cpu_all, linf_all, oracle_cache = {}, {}, {}

    def cpu_leg(name, workload, device_grads):
        if workload not in oracle_cache:
            oracle_cache.clear()            # (one workload's oracle gradients at a time: cfg3's sample alone is 80 MB)
            c, og = cpu_baseline(workload, args.size)
            cpu_all[workload] = c
            oracle_cache[workload] = og
        if device_grads is not None:
            linf_all[name] = grad_linf_rel(device_grads, oracle_cache[workload])

    head_name = args.workload + ("_lazy" if args.lazy else "")
    if check_grads:
        cpu_leg(head_name, args.workload, head_grads)
    head_grads = None

    # ---- the other BASELINE configs, same process, after the headline's timed region -------------------
    secondary = None
    if not args.no_secondary and not args.graph and (not args.size or args.force_secondary or os.environ.get("MDHIP_BENCH_HOST_COMM") == "1"):
        secondary = {}
        gc.collect()
        lib.empty_cache()
        k_sec, w_sec = min(args.steps, 10), min(args.warmup, 2)
        if use_dist:
            plan = [("cfg2_weak", "cfg2", False)] if args.workload == "cfg4" else [("cfg4_strong", "cfg4", False)]
        else:
            plan = [(f"{wl}{'_lazy' if lz else ''}", wl, lz) for wl in ("cfg3", "cfg4", "cfg5", "cfg2") for lz in (False, True)
                    if not (wl in ("cfg5", "cfg2") and lz)]
            plan = [p for p in plan if not (p[1] == args.workload and p[2] == bool(args.lazy))]
            # cfg5's sweep is eleven short launches (five 124-us GEMMs, fills, sums): the same sweep replayed from ONE captured
            # hipGraph (graph.CapturedSweep, the mode SweepCache serves reuse_graph() with) — what the launch boundaries cost
            from minidiff_amd.graph import can_capture
            if args.workload != "cfg5" and can_capture(lib):      # (the CPU test double cannot replay)
                plan.insert(plan.index(("cfg5", "cfg5", False)) + 1, ("cfg5_graph", "cfg5", "graph"))
        keep, keep_wl = None, None
        for name, wl, lz in plan:
            if wl != keep_wl:
                keep, keep_wl = {}, wl
                gc.collect()
                lib.empty_cache()
            try:
                # a timed region of >= ~40 ms for the short sweeps too (ten sweeps of cfg3 --lazy are 4 ms: the first ones after
                # the workload switch weighed 5-7 %); explicit small --steps (tests) are kept
                k_this = k_sec
                if args.steps >= 10 and not args.size:
                    k_this = max(k_sec, SECONDARY_STEPS.get(name, k_sec))
                if lz == "graph":
                    r = run(wl, False, k_this, w_sec, graph=True, size=args.size, keep=None)
                else:
                    r = run(wl, lz, k_this, w_sec, size=args.size, keep=keep, want_grads=check_grads)
                if use_dist:    # its own single-GPU figure, same job
                    keep.clear()
                    r["single_gpu_same_workload"] = solo_run(wl, lz)
                grads = r.pop("_grads", None)
                if check_grads:
                    cpu_leg(name, wl, grads)
                grads = None
                secondary[name] = r
            except Exception as e:  # a secondary config must never take the headline down
                # (N > 1 too: a failure that every rank hits at the same point — the deterministic kind — is recorded and the
                # headline line still goes out; a failure on SOME ranks would stall the others in a collective either way)
                secondary[name] = {"error": f"{type(e).__name__}: {e}"}
                print(f"[bench] secondary {name} failed: {type(e).__name__}: {e}", file=sys.stderr)
        keep = None
        oracle_cache.clear()
        gc.collect()

    nd.DeviceArray._materialize = _plain_materialize
    if rank == 0:
        results = {head_name: head}
        results.update(secondary or {})
        flat = flat_keys({k: v for k, v in results.items() if k != head_name or head_name != "cfg2"})
        linf_max = max((v for d in linf_all.values() for k, v in d.items() if not k.endswith("_raw") and not k.startswith("mask_")), default=None)
        roof = dict(head["roofline"] or {})
        roof = {k: v for k, v in roof.items() if not isinstance(v, (dict, list))}    # scalars only: what the driver's record keeps
        if roof.get("traffic_committed_source"):
            roof["traffic_committed_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc passes)"
        for k in ("note", "algorithmic_bytes_per_launch"):    # (derivable: 3 x 4 N^2; in the detail file)
            roof.pop(k, None)
        if roof.get("traffic_committed") is None:
            roof.pop("traffic_committed", None)
            roof.pop("traffic_committed_source", None)
        roof.update(flat)
        if linf_max is not None:
            roof["grad_linf_rel_max"] = linf_max
        for wl, c in cpu_all.items():
            roof[f"cpu_{wl}_passes_per_s"] = c["value"]
        if "mask_flips" in linf_all.get("cfg4", {}):    # relu decisions that differ from the NumPy engine's (of batch x 4096)
            roof["cfg4_mask_flips"] = linf_all["cfg4"]["mask_flips"]
            roof["cfg4_grad_linf_rel_raw"] = max(linf_all["cfg4"].get("W_raw", 0.0), linf_all["cfg4"].get("b_raw", 0.0))
        cpu = None
        if args.workload in cpu_all:
            cpu = dict(cpu_all[args.workload])
            for wl, c in cpu_all.items():      # the NumPy engine on every config (cfg3: N = 1e7 sample, scaled linearly)
                if wl != args.workload:
                    cpu[f"{wl}_value"] = c["value"]
        cfg = {k: v for k, v in head["config"].items() if not isinstance(v, (dict, list)) and not (k.startswith("sweep_trial_") and v is None)}
        if not use_dist:   # (one GPU: no collective — the all-reduce fields would only say so)
            cfg = {k: v for k, v in cfg.items() if not k.startswith("allreduce_") and k != "rccl_ranks"}
        if isinstance(head["config"].get("graph_replay"), dict):
            cfg["graph_replay"] = True
            cfg["graph_segments"] = head["config"]["graph_replay"]["segments"]
            cfg["graph_collective_calls"] = head["config"]["graph_replay"]["collective_calls"]
        line = {
            "metric": "forward+backward passes/sec on 4096x4096 fp32 matmul+elementwise graph",
            "value": head["value"], "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": cfg, "roofline": roof, "cpu_baseline": cpu,
            # the metric's second half: max over every gradient of every config run, norm-wise against the NumPy engine
            # on the same seeds (bar 1e-5 for fp32). None when the CPU leg is skipped (N > 1, --no-cpu-baseline).
            "grad_linf_rel_max": linf_max,
        }
        if "tensors_per_s" in head:
            line["tensors_per_s"] = head["tensors_per_s"]
        if solo is not None:
            line["single_gpu_value"] = solo["value"]
            if "tensors_per_s" in solo:
                line["single_gpu_tensors_per_s"] = solo["tensors_per_s"]
        detail = {"line": line, "head": head, "secondary": secondary, "cpu_baseline": cpu_all, "grad_linf_rel": linf_all,
                  "single_gpu_same_workload": solo, "preroll_sweeps": head["preroll_sweeps"], "single_sweep_ms": head["single_sweep_ms"],
                  # per-kernel durations are the kernels' own dispatch timestamps; a marker bracket (only where a call launches
                  # no attachable kernel) costs this much stream time
                  "event_bracket_overhead_ms": event_overhead_ms}
        # ---- the stdout line: < 2 KB (the driver keeps the last 2 KB of stdout and the scalars of roofline / config / cpu_baseline)
        def rounded(o):
            if isinstance(o, dict):
                return {k: rounded(v) for k, v in o.items()}
            return _r(o)
        out = rounded(line)
        drop = list(DROP_ORDER)
        text = json.dumps(out, separators=(",", ":"))
        while len(text) > 1950 and drop:
            out["roofline"].pop(drop.pop(0), None)
            text = json.dumps(out, separators=(",", ":"))
        try:
            with open(args.detail, "w") as fh:
                json.dump(detail, fh, indent=1, default=str)
        except OSError as e:
            print(f"[bench] could not write {args.detail}: {e}", file=sys.stderr)
        print("[bench-detail] " + json.dumps(detail, default=str), file=sys.stderr)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(text)
        sys.stdout.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
