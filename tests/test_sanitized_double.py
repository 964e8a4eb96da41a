"""The CPU test double rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`) runs a slice of the
CPU suite once: the double shares md_ops.h / md_dispatch.h / md_vm.h / md_common.h (per-element semantics, dispatch, iteration
spaces, the fused-program interpreter) with the product, so an out-of-bounds walk or undefined behaviour in those headers is
caught here — GPU sanitizers are not available on this pool. The C route of eager calls (csrc/fastpath.c: reference counts, GC
hooks, descriptors on the stack) is rebuilt with the same sanitizers and loaded in place of the product build, so the slice —
its own tests included — runs through it. One subprocess: the sanitizer runtime has to be preloaded into the interpreter."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SLICE = ["test_shim.py", "test_large_shape_paths.py", "test_lazy_fusion.py", "test_golden_device.py", "test_fuzz_differential.py",
         "test_device_rng.py", "test_ops_reference_style.py", "test_fastpath.py", "test_std_fused.py", "test_graph.py", "test_narrow_dtypes.py", "test_api_differential.py", "test_training_loop.py"]


def test_cpu_suite_slice_on_the_sanitized_double(on_gpu):
    if on_gpu:
        pytest.skip("CPU-double check")
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan_rt, MDHIP_HOST_DOUBLE=os.path.join(ROOT, "oracle", "_build", "libmdhip_host_asan.so"),
               MDHIP_FORCE_HOST="1", MDHIP_FASTPATH_SO=os.path.join(ROOT, "oracle", "_build", "_fastpath_asan.so"), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "-p", "no:xdist"]
                       + [os.path.join(HERE, f) for f in SLICE], env=env, capture_output=True, text=True, timeout=1500)
    tail = p.stdout[-3000:] + p.stderr[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
