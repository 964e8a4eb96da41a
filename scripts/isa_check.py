#!/usr/bin/env python3
"""Facts about the headline GEMM kernels read from the shipped gfx950 code object (no GPU needed): what
tests/test_isa_guard.py asserts so that a toolchain bump cannot silently undo the three compiler-steering devices the
94 % of fp32 MFMA peak depends on.

    isa_check.py [OBJ]      OBJ: minidiff_amd/libmdhip.so (default: the SHIPPED library) or build/mdhip/gemm.o — any object
                            or shared library; every gfx950 offload bundle in it is searched for the kernels

Per kernel: MFMA count, LDS-DMA count, DMAs of the MAIN LOOP by address form (`vN, s[base]` scalar-base form vs the
64-bit `v[N:N+1], off` form), v_lshl_add_u64 in the main loop, ds_write count, scratch use, and whether a `s_waitcnt
vmcnt(0)` sits between the k-tile barrier and the first fragment read behind it."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(obj, wanted=()):
    """-> `llvm-objdump -d --mcpu=gfx950` text of the kernels of `obj` whose mangled names contain one of `wanted` (all
    kernels when empty). `obj` may hold SEVERAL gfx950 offload bundles (a shared library has one per translation unit):
    every bundle is unpacked, its symbol table searched, and only the matching symbols are disassembled."""
    tmp = tempfile.mkdtemp(prefix="mdhip_isa_")
    try:
        local = os.path.join(tmp, "in.o")
        shutil.copy(obj, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        bundles = sorted(f for f in os.listdir(tmp) if "gfx950" in f)
        if not bundles:
            raise RuntimeError(f"no gfx950 bundle in {obj}")
        text = []
        for co in bundles:
            path = os.path.join(tmp, co)
            if not wanted:
                text.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", path], check=True, capture_output=True, text=True).stdout)
                continue
            syms = subprocess.run([f"{LLVM}/llvm-objdump", "-t", path], check=True, capture_output=True, text=True).stdout
            names = sorted({ln.split()[-1] for ln in syms.splitlines() if " F " in ln and any(w in ln for w in wanted)})
            names = [n for n in names if not n.endswith(".kd")]
            if names:
                text.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", "--disassemble-symbols=" + ",".join(names), path],
                                           check=True, capture_output=True, text=True).stdout)
        return "\n".join(text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def analyse(name, raw_lines):
    """raw_lines: the kernel's lines WITH addresses (as objdump prints them)."""
    ins, addr = [], []
    for ln in raw_lines:
        body, _, tail = ln.partition("//")
        a = re.search(r"([0-9A-Fa-f]{12}):", tail)
        ins.append(body.strip())
        addr.append(int(a.group(1), 16) if a else None)
    # backward branches: s_cbranch_* with a negative 16-bit word offset
    loops = []
    for i, s in enumerate(ins):
        m = re.match(r"s_cbranch_\w+\s+(\d+)", s)
        if m and addr[i] is not None:
            off = int(m.group(1))
            if off >= 0x8000:
                target = addr[i] + 4 + (off - 0x10000) * 4
                j = next((k for k in range(i, -1, -1) if addr[k] == target), None)
                if j is not None:
                    loops.append((j, i))
    best = max(loops, key=lambda ab: sum(1 for s in ins[ab[0]:ab[1]] if s.startswith("v_mfma")), default=None)
    body = ins[best[0]:best[1] + 1] if best else []
    dma = [s for s in body if s.startswith("global_load_lds")]
    saddr = [s for s in dma if re.search(r"global_load_lds_dwordx4\s+v\d+,\s*s\[", s)]
    vaddr = [s for s in dma if re.search(r"global_load_lds_dwordx4\s+v\[\d+:\d+\],\s*off", s)]
    # a vmcnt(0) drain between a barrier and the first LDS read behind it
    drain = 0
    for i, s in enumerate(body):
        if s.startswith("s_barrier"):
            for t in body[i + 1:]:
                if t.startswith("ds_read"):
                    break
                if t.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", t):
                    drain += 1
                    break
    return {
        "mfma_total": sum(1 for s in ins if s.startswith("v_mfma")),
        "mfma_loop": sum(1 for s in body if s.startswith("v_mfma")),
        "dma_total": sum(1 for s in ins if s.startswith("global_load_lds")),
        "dma_loop": len(dma), "dma_loop_saddr": len(saddr), "dma_loop_vaddr64": len(vaddr),
        "lshl_add_u64_loop": sum(1 for s in body if s.startswith("v_lshl_add_u64")),
        "valu_loop_non_mfma": sum(1 for s in body if s.startswith("v_") and not s.startswith("v_mfma")),
        "ds_write": sum(1 for s in ins if s.startswith("ds_write")),
        "scratch": sum(1 for s in ins if s.startswith("scratch_") or "buffer_store" in s and "offen" in s and "s[0:3]" in s),
        "vmcnt0_between_barrier_and_first_read": drain,
        "loop_instructions": len(body),
    }


def raw_kernels(text):
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
        elif cur is not None and ln.startswith("\t"):
            cur.append(ln)
    return out


HEADLINE = {
    "NN 256x256x32": "k_gemm_f32_kc_gldsILi256ELi256ELi32ELi2ELi2ELb0ELi0ELi0ELi2ELb0EE",
    "NT 256x256x32": "k_gemm_f32_kc_gldsILi256ELi256ELi32ELi2ELi2ELb1ELi0ELi0ELi2ELb0EE",
    "TN 256x256x32": "k_gemm_f32_tn_gldsILi256ELi256ELi32ELi2ELi2ELi0ELi2EE",
}


def report(obj=None):
    obj = obj or os.path.join(ROOT, "minidiff_amd", "libmdhip.so")
    ks = raw_kernels(disassemble(obj, tuple(HEADLINE.values())))
    res = {}
    for label, frag in HEADLINE.items():
        names = [n for n in ks if frag in n]
        if len(names) != 1:
            raise RuntimeError(f"{label}: {len(names)} kernels match {frag}")
        res[label] = analyse(names[0], ks[names[0]])
    return res


if __name__ == "__main__":
    r = report(sys.argv[1] if len(sys.argv) > 1 else None)
    for label, d in r.items():
        print(label, d)
