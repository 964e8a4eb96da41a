"""Guard on the three headline GEMM kernels (cfg2's NN / NT / TN at 256 x 256 x 32), read from the SHIPPED library
minidiff_amd/libmdhip.so — the file that travels to the GPU box and that the product loads — no GPU needed (llvm-objdump of
every gfx950 offload bundle in it: scripts/isa_check.py). Runs in the CPU suite and, the library being in-tree, on the GPU box too.

The 94-95 % of fp32 MFMA peak depends on compiler behaviour that nothing else pins: the direct-to-LDS DMAs must stay in
the `vN, s[base:base+1]` address form (scalar base + 32-bit lane offset: no vector ALU work per DMA), the two LDS buffers
must stay separate objects (no `s_waitcnt vmcnt(0)` drain between the k-tile barrier and the first fragment read), no
staging through registers (`ds_write`), no scratch. A toolchain bump that silently brings back the 64-bit per-lane form
fails here instead of costing 3 % on the GPU box unnoticed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
OBJ = os.path.join(ROOT, "minidiff_amd", "libmdhip.so")


@pytest.fixture(scope="module")
def isa():
    if not os.path.exists(OBJ):
        pytest.skip("minidiff_amd/libmdhip.so not built (run __graft_entry__.build())")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    import isa_check
    return isa_check.report(OBJ)


def _check(k):
    # 256 x 256 x 32 tile, four waves: 4 x 4 fragments x 16 k-pairs... = 256 MFMAs per k-tile; the loop holds two k-tiles,
    # the tail one more
    assert k["mfma_total"] == 768 and k["mfma_loop"] == 512, k
    assert k["dma_total"] == 64 and k["dma_loop"] == 32, k
    # every main-loop DMA in the scalar-base form, none in the 64-bit per-lane form, no 64-bit vector adds feeding them
    assert k["dma_loop_saddr"] == 32 and k["dma_loop_vaddr64"] == 0 and k["lshl_add_u64_loop"] == 0, k
    # what is left of vector ALU work in two k-tiles (512 MFMAs): a handful of moves
    assert k["valu_loop_non_mfma"] <= 16, k
    assert k["ds_write"] == 0 and k["scratch"] == 0, k
    assert k["vmcnt0_between_barrier_and_first_read"] == 0, k


LAYOUTS = ["NN 256x256x32", "NT 256x256x32", "TN 256x256x32"]


@pytest.mark.parametrize("layout", LAYOUTS)
def test_headline_gemm_kernel_shape(isa, layout):
    _check(isa[layout])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", LAYOUTS)
def test_headline_gemm_kernel_shape_of_the_library_on_the_gpu_box(isa, layout):
    """The same facts in the driver's `-m gpu` run: what is checked there is the very file that run loads."""
    _check(isa[layout])
