#!/usr/bin/env python3
"""bench.py — forward+backward passes/sec of the minidiff hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|cfg4|cfg5]

Metric (BASELINE.json): forward+backward passes/sec on the 4096x4096 fp32 matmul
graph. Default workload = configs[1] (cfg2): C = A @ B; C.backward() — 3 GEMMs
(NN, NT, TN), 412,316,860,416 FLOP per sweep, inputs resident in HBM.
With N > 1 the sweep is batch-sharded: every rank owns its own 4096-row batch
block A_r (rows are independent), B is the replicated parameter, and B.grad is
summed with ONE RCCL all-reduce (67,108,864 B) per sweep; per-GPU work is fixed
("weak" scaling) and `value` counts the sweeps all ranks completed per second.
`--workload cfg4` runs BASELINE's MLP config instead (global batch 8192 split
over the ranks, bucketed [W.grad || b.grad] all-reduce).

One JSON line on stdout (rank 0). `roofline` prices the dominant kernel from HIP
events recorded on the library's stream inside the timed region; `cpu_baseline`
times the NumPy oracle (the reference's arithmetic) on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix, dense
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--size", type=int, default=0, help="override the problem size (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lazy", action="store_true",
                    help="opt-in lazy fusion of elementwise chains (minidiff_amd/lazy.py); default is eager")
    ap.add_argument("--graph", action="store_true",
                    help="capture one sweep into a hipGraph after warm-up and time K replays of it (N=1 only)")
    ap.add_argument("--comm", default=os.environ.get("MDHIP_COMM", "rccl"), choices=["rccl", "torch"])
    return ap.parse_args()


class KernelTimer:
    """Brackets chosen backend calls with HIP events on libmdhip's stream."""

    def __init__(self, lib, capacity=4096):
        import ctypes as C
        self.C, self.lib = C, lib
        self.pool, self.used, self.enabled = [], [], False
        self.capacity = capacity

    def _event(self):
        if self.pool:
            return self.pool.pop()
        ev = self.C.c_void_p()
        self.lib.event_create(self.C.byref(ev))
        return ev

    def wrap(self, fn, tag):
        def timed(*a, **kw):
            if not self.enabled or len(self.used) >= self.capacity:
                return fn(*a, **kw)
            e0, e1 = self._event(), self._event()
            self.lib.event_record(e0)
            out = fn(*a, **kw)
            self.lib.event_record(e1)
            self.used.append((tag, e0, e1))
            return out
        return timed

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
    vals = [v["hbm_bytes_per_launch"] for k, v in pmc.get("kernels", {}).items() if k.startswith(prefix)]
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
        sample = "full cfg4 sweep (global batch on one host), best of 3 after 1 warm-up"
        reps = 3
    _, step = workloads.MAKERS[workload](md, **kw)
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
    return {
        "value": scale / best, "unit": "passes/s", "cores": cores, "kind": "port",
        "sample": sample + f"; numpy {np.__version__}; GEMM threads = all cores, ufuncs single-threaded",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MDHIP_DEVICE"] = str(local_rank)

    # MDHIP_BENCH_FORCE_DIST=1 runs the N>1 code path (process group, ncclUniqueId exchange, RCCL
    # communicator, per-sweep all-reduce, max-over-ranks timing) at world size 1 — the only way to
    # rehearse it on a one-GPU box
    force_dist = os.environ.get("MDHIP_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
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
    from minidiff_amd.hip_backend import HipBackendTable
    from minidiff_amd.tape import build_engine

    lib = _capi.load()  # ImportError if the HIP extension is missing: no fallback
    timer = KernelTimer(lib)
    if args.lazy:
        from minidiff_amd import ndarray as _nd
        _nd.set_lazy(True)

    # instrumented copy of the table: same functions, dominant kernels bracketed by events
    dominant = {"cfg2": ["matmul"], "cfg4": ["matmul"], "cfg5": ["matmul"],
                "cfg3": ["sin", "cos", "multiply", "power", "sum"]}[args.workload]
    ns = {k: v for k, v in vars(HipBackendTable).items() if not k.startswith("__")}
    for name in dominant:
        ns[name] = staticmethod(timer.wrap(getattr(HipBackendTable, name), name))
    Table = type("HipBackendTableTimed", (), ns)
    md = build_engine(Table, "hip")

    kw = {}
    if args.workload == "cfg2":
        kw = {"rank": rank}
        if args.size:
            kw["n"] = args.size
    elif args.workload == "cfg4":
        kw = {"rank": rank, "world": world}
        if args.size:
            kw["batch"] = args.size
    elif args.size:
        kw = {"n": args.size}
    state, step = workloads.MAKERS[args.workload](md, **kw)

    comm = None
    comm_kind = "none"
    if use_dist and args.workload in ("cfg2", "cfg4"):
        if args.comm == "rccl":
            err = None
            try:
                comm = dp.RcclComm(rank, world, dist)
            except Exception as e:  # communicator could not be built on this rank
                err = e
            # every rank must take the same path: agree over the gloo control plane
            flags = [None] * world
            dist.all_gather_object(flags, err is None)
            if all(flags):
                comm_kind = "rccl-direct"
            else:
                if comm is not None:
                    comm.close()
                    comm = None
                print(f"[rank {rank}] direct RCCL communicator unavailable ({err or 'failed on another rank'}); "
                      "using torch.distributed nccl", file=sys.stderr)
        if comm is None:  # same data path through torch's RCCL
            comm = dp.TorchComm(rank, world, dist, torch)
            comm_kind = "rccl-torch"
    sync = dp.GradSync(md, state["params"] if args.workload == "cfg4" else state["params"][:1], comm, force=force_dist,
                       overlap=os.environ.get("MDHIP_DP_OVERLAP", "1") != "0")

    def sweep():
        step()
        sync()

    def barrier():
        lib.sync()
        if use_dist:
            dist.barrier()

    # untimed pre-roll before the W warm-up sweeps: the first ~15 ms of work after process start run
    # at ramping clocks and grow the allocator cache (W <= 2 alone measured 7 % low on cfg2)
    # (a fixed count, not a time: every rank must issue the same number of collectives)
    preroll = {"cfg2": 12, "cfg3": 20, "cfg4": 10, "cfg5": 40}[args.workload]
    for _ in range(preroll):
        sweep()
    lib.sync()

    captured = None
    if args.graph:
        if use_dist:
            raise SystemExit("--graph is a single-GPU mode (the gradient all-reduce is not captured)")
        # single kernels cannot be bracketed inside a replay: the per-kernel durations the roofline
        # needs come from the (eager, identical) warm-up sweeps instead
        sweep()  # cold start (code object load, allocator growth) stays out of the kernel averages
        timer.enabled = True
        for _ in range(max(args.warmup, 1)):
            sweep()
        lib.sync()
        timer.enabled = False
        from minidiff_amd.graph import CapturedSweep
        captured = CapturedSweep(step, warmup=0)
        captured.replay()
        run_one = captured.replay
    else:
        for _ in range(args.warmup):
            sweep()
        run_one = sweep
    barrier()
    timer.enabled = not args.graph
    t0 = time.perf_counter()
    for _ in range(args.steps):
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
    ms_per_step = elapsed / args.steps * 1e3

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
    if args.workload == "cfg4":
        value = args.steps / elapsed
        scaling = "strong"
    else:
        value = world * args.steps / elapsed
        scaling = "weak"

    roofline = None
    pmc = load_pmc_traffic()
    if args.workload in ("cfg2", "cfg4", "cfg5"):
        durs = kernel_ms.get("matmul", [])
        n_gemm = {"cfg2": 3, "cfg4": 2, "cfg5": 5}[args.workload]
        flop_per_launch = state["flops"] / n_gemm
        if durs:
            avg = sum(durs) / len(durs)
            ach = flop_per_launch / (avg * 1e-3) / 1e12
            traffic = pmc_mean(pmc, "k_gemm_f32_mfma") if args.workload == "cfg2" and not args.size else None
            roofline = {"bound": "mfma", "kernel": "k_gemm_f32_mfma", "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                        "traffic_source": pmc.get("_source") if traffic else None,
                        "launches": len(durs), "avg_launch_ms": avg, "flop_per_launch": flop_per_launch,
                        "algorithmic_bytes_per_launch": 3 * 4 * (args.size or 4096) ** 2 if args.workload == "cfg2" else None}
    elif args.lazy:
        # fused: one reduce pass over (x, y) for the loss (8N), one two-output pass for both
        # gradients (reads x, y once, writes x.grad and y.grad: 16N) = SURVEY 8d's fused lower bound
        n = state["rows"]
        fused_bytes = 24 * n
        ach = fused_bytes / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_fused (reduce) + k_fused (two-output eval), run-time specialised",
                    "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                    "fused_algorithmic_bytes_per_sweep": fused_bytes, "eager_algorithmic_bytes_per_sweep": state["bytes"],
                    "note": "achieved = fused bytes / whole-sweep time (3 launches incl. the reduction's finish); the eager figure (100N bytes) is not mixed in"}
    else:
        # dominant kernel of the eager chain: the f32 multiply (6 of the 11 launches per sweep:
        # five read 2 x 4N and write 4N, the scaled stride-0 seed only writes 4N -> 64N bytes per sweep)
        n = state["rows"]
        durs = kernel_ms.get("multiply", [])
        if durs:
            avg = sum(durs) / len(durs)
            bytes_per_launch = 64 * n / 6
            ach = bytes_per_launch / (avg * 1e-3) / 1e9
            tot_ms = sum(sum(v) for v in kernel_ms.values())
            traffic = pmc_mean(pmc, "k_binary_fast<BMul, float, float, float, float") if not args.size else None
            roofline = {"bound": "hbm", "kernel": "k_binary_fast<BMul,f32>", "achieved": ach, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                        "traffic_source": pmc.get("_source") if traffic else None,
                        "launches": len(durs), "avg_launch_ms": avg, "algorithmic_bytes_per_launch": bytes_per_launch,
                        "whole_sweep": {"algorithmic_bytes": state["bytes"], "kernel_ms": tot_ms / args.steps,
                                        "GB/s": state["bytes"] * args.steps / (tot_ms * 1e-3) / 1e9,
                                        "per_kernel_ms": {k: sum(v) / len(v) for k, v in kernel_ms.items()}}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, args.size)

    if comm is not None:
        comm.close()
    if captured is not None:
        captured.close()
    if rank == 0:
        n = args.size or {"cfg2": 4096, "cfg3": 100_000_000, "cfg4": 8192, "cfg5": 2048}[args.workload]
        line = {
            "metric": "forward+backward passes/sec on 4096x4096 fp32 matmul+elementwise graph",
            "value": value, "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preroll_sweeps": preroll,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": {
                "cfg2": f"cfg2: C=A@B ({n}x{n} fp32), C.backward(); 3 GEMMs NN/NT/TN; per-rank batch block, B.grad all-reduced",
                "cfg3": f"cfg3: sum((sin(x)*y)**2).backward(), N={n} fp32" + (" [lazy fusion]" if args.lazy else " [eager: 11 kernels]"),
                "cfg4": f"cfg4: sum(relu(X@W+b)).backward(), global batch {n} x 4096 -> 4096, row-sharded",
                "cfg5": f"cfg5: second order on {n}x{n} matmul, 5 GEMMs"}[args.workload],
                "parallelism": f"dp{world}", "graph_replay": bool(args.graph), "collective": comm_kind, "allreduce_bytes": sync.nbytes if use_dist else 0,
                "allreduce_overlapped_sweeps": sync.overlapped,
                "allreduce_alone_ms": allreduce_ms, "allreduce_busbw_GBps": busbw},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
